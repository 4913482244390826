// Host-side pairing for batch verification of Groth16 proofs (SURVEY.md 8f rank 4:
// `SNARK::verify_with_processed_vk`, /root/reference/snark/src/lib.rs:76-80; upstream ark-groth16 verifier.rs).
//
// What runs where: the random linear combination of a batch is group arithmetic in G1 -- the two multi-scalar sums go
// through the device MSM like every other MSM of this library -- while the k + 3 Miller loops and the ONE final
// exponentiation of a batch of k proofs are sequential tower-field arithmetic with no width to fill a GPU; they run on host
// threads in the library's own host-compiled field code.  Verification is not on the hot path of the prover; this exists
// so that a host that proves on the device can also check what it produced without leaving the C ABI.
//
// Construction (textbook, kept simple): F_q2 = F_q[u]/(u^2 + 1), F_q6 = F_q2[v]/(v^3 - xi), F_q12 = F_q6[w]/(w^2 - v)
// with xi = 1 + u (BLS12-381, M-type twist) or 9 + u (BN254, D-type twist); affine Miller loop over the twist with the
// line evaluated sparsely into F_q12; BLS12-381: ate loop over |x| = 0xd201000000010000 and a conjugation for x < 0;
// BN254: optimal ate over 6x + 2 with the two Frobenius steps; final exponentiation = (conj(f) / f)^((q^6 + 1) / r) with
// the exponent computed once by long division.
#pragma once
#include <thread>
#include <vector>
#include "common.h"

namespace ark355 {

// ---- small unsigned big integers (exponents only) -------------------------------------------------------------------
struct BigU {
  std::vector<uint32_t> l;      // little-endian limbs, no leading zeros
  void trim() {
    while (!l.empty() && l.back() == 0) l.pop_back();
  }
  static BigU from_u32(const uint32_t* p, int n) {
    BigU r;
    r.l.assign(p, p + n);
    r.trim();
    return r;
  }
  static BigU small(uint32_t v) {
    BigU r;
    if (v) r.l.push_back(v);
    return r;
  }
  size_t bits() const {
    if (l.empty()) return 0;
    size_t b = 32 * (l.size() - 1);
    uint32_t t = l.back();
    while (t) {
      b++;
      t >>= 1;
    }
    return b;
  }
  bool bit(size_t i) const { return (i >> 5) < l.size() && ((l[i >> 5] >> (i & 31)) & 1u); }
  static BigU mul(const BigU& a, const BigU& b) {
    BigU r;
    r.l.assign(a.l.size() + b.l.size() + 1, 0);
    for (size_t i = 0; i < a.l.size(); i++) {
      uint64_t c = 0;
      for (size_t j = 0; j < b.l.size(); j++) {
        c += (uint64_t)a.l[i] * b.l[j] + r.l[i + j];
        r.l[i + j] = (uint32_t)c;
        c >>= 32;
      }
      for (size_t k = i + b.l.size(); c; k++) {
        c += r.l[k];
        r.l[k] = (uint32_t)c;
        c >>= 32;
      }
    }
    r.trim();
    return r;
  }
  static BigU add(const BigU& a, const BigU& b) {
    BigU r;
    uint64_t c = 0;
    for (size_t i = 0; i < std::max(a.l.size(), b.l.size()) || c; i++) {
      c += (i < a.l.size() ? a.l[i] : 0u);
      c += (i < b.l.size() ? b.l[i] : 0u);
      r.l.push_back((uint32_t)c);
      c >>= 32;
    }
    r.trim();
    return r;
  }
  static int cmp(const BigU& a, const BigU& b) {
    if (a.l.size() != b.l.size()) return a.l.size() < b.l.size() ? -1 : 1;
    for (size_t i = a.l.size(); i-- > 0;)
      if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
    return 0;
  }
  static BigU sub(const BigU& a, const BigU& b) {      // a >= b
    BigU r = a;
    int64_t br = 0;
    for (size_t i = 0; i < r.l.size(); i++) {
      int64_t t = (int64_t)r.l[i] - (i < b.l.size() ? b.l[i] : 0u) + br;
      r.l[i] = (uint32_t)t;
      br = t >> 32;
    }
    r.trim();
    return r;
  }
  // floor(a / b), schoolbook shift-subtract (runs once per curve)
  static BigU div(const BigU& a, const BigU& b, BigU* rem_out = nullptr) {
    BigU q, rem;
    q.l.assign(a.l.size(), 0);
    for (size_t i = a.bits(); i-- > 0;) {
      // rem = rem * 2 + bit
      uint32_t c = a.bit(i) ? 1u : 0u;
      for (size_t k = 0; k < rem.l.size(); k++) {
        const uint32_t n = (rem.l[k] << 1) | c;
        c = rem.l[k] >> 31;
        rem.l[k] = n;
      }
      if (c) rem.l.push_back(c);
      if (cmp(rem, b) >= 0) {
        rem = sub(rem, b);
        q.l[i >> 5] |= 1u << (i & 31);
      }
    }
    q.trim();
    if (rem_out) *rem_out = rem;
    return q;
  }
};

template <class Curve>
struct PairingHost {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using Fr = typename Curve::Fr;
  static constexpr bool BN = Curve::ID == ARK355_BN254;

  static Fq2 mul_fq(const Fq2& a, const Fq& k) { return Fq2{Fq::mul(a.c0, k), Fq::mul(a.c1, k)}; }
  static Fq2 conj2(const Fq2& a) { return Fq2{a.c0, Fq::neg(a.c1)}; }
  // a * xi, xi = 1 + u (BLS12-381) / 9 + u (BN254)
  static Fq2 mul_xi(const Fq2& a) {
    if (BN) {
      const Fq a0_8 = Fq::dbl(Fq::dbl(Fq::dbl(a.c0))), a1_8 = Fq::dbl(Fq::dbl(Fq::dbl(a.c1)));
      return Fq2{Fq::sub(Fq::add(a0_8, a.c0), a.c1), Fq::add(Fq::add(a1_8, a.c1), a.c0)};
    }
    return Fq2{Fq::sub(a.c0, a.c1), Fq::add(a.c0, a.c1)};
  }

  struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 add(const Fq6& a, const Fq6& b) { return Fq6{Fq2::add(a.c0, b.c0), Fq2::add(a.c1, b.c1), Fq2::add(a.c2, b.c2)}; }
    static Fq6 sub(const Fq6& a, const Fq6& b) { return Fq6{Fq2::sub(a.c0, b.c0), Fq2::sub(a.c1, b.c1), Fq2::sub(a.c2, b.c2)}; }
    static Fq6 neg(const Fq6& a) { return Fq6{Fq2::neg(a.c0), Fq2::neg(a.c1), Fq2::neg(a.c2)}; }
    static Fq6 mul_v(const Fq6& a) { return Fq6{mul_xi(a.c2), a.c0, a.c1}; }
    static Fq6 mul(const Fq6& a, const Fq6& b) {
      const Fq2 t0 = Fq2::mul(a.c0, b.c0), t1 = Fq2::mul(a.c1, b.c1), t2 = Fq2::mul(a.c2, b.c2);
      const Fq2 m12 = Fq2::mul(Fq2::add(a.c1, a.c2), Fq2::add(b.c1, b.c2));
      const Fq2 m01 = Fq2::mul(Fq2::add(a.c0, a.c1), Fq2::add(b.c0, b.c1));
      const Fq2 m02 = Fq2::mul(Fq2::add(a.c0, a.c2), Fq2::add(b.c0, b.c2));
      return Fq6{Fq2::add(t0, mul_xi(Fq2::sub(Fq2::sub(m12, t1), t2))),
                 Fq2::add(Fq2::sub(Fq2::sub(m01, t0), t1), mul_xi(t2)),
                 Fq2::add(Fq2::sub(Fq2::sub(m02, t0), t2), t1)};
    }
    static Fq6 inv(const Fq6& a) {
      const Fq2 c0 = Fq2::sub(Fq2::sqr(a.c0), mul_xi(Fq2::mul(a.c1, a.c2)));
      const Fq2 c1 = Fq2::sub(mul_xi(Fq2::sqr(a.c2)), Fq2::mul(a.c0, a.c1));
      const Fq2 c2 = Fq2::sub(Fq2::sqr(a.c1), Fq2::mul(a.c0, a.c2));
      const Fq2 t = Fq2::add(Fq2::mul(a.c0, c0), mul_xi(Fq2::add(Fq2::mul(a.c2, c1), Fq2::mul(a.c1, c2))));
      const Fq2 ti = Fq2::inv(t);
      return Fq6{Fq2::mul(c0, ti), Fq2::mul(c1, ti), Fq2::mul(c2, ti)};
    }
    bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
  };

  struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
    static Fq12 mul(const Fq12& a, const Fq12& b) {
      const Fq6 t0 = Fq6::mul(a.c0, b.c0), t1 = Fq6::mul(a.c1, b.c1);
      const Fq6 m = Fq6::mul(Fq6::add(a.c0, a.c1), Fq6::add(b.c0, b.c1));
      return Fq12{Fq6::add(t0, Fq6::mul_v(t1)), Fq6::sub(Fq6::sub(m, t0), t1)};
    }
    static Fq12 sqr(const Fq12& a) { return mul(a, a); }
    static Fq12 conj(const Fq12& a) { return Fq12{a.c0, Fq6::neg(a.c1)}; }
    static Fq12 inv(const Fq12& a) {
      const Fq6 t = Fq6::sub(Fq6::mul(a.c0, a.c0), Fq6::mul_v(Fq6::mul(a.c1, a.c1)));
      const Fq6 ti = Fq6::inv(t);
      return Fq12{Fq6::mul(a.c0, ti), Fq6::neg(Fq6::mul(a.c1, ti))};
    }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    static Fq12 pow(const Fq12& a, const BigU& e) {
      Fq12 r = one();
      for (size_t i = e.bits(); i-- > 0;) {
        r = sqr(r);
        if (e.bit(i)) r = mul(r, a);
      }
      return r;
    }
  };

  static Fq2 pow2(const Fq2& a, const BigU& e) {
    Fq2 r = Fq2::one();
    for (size_t i = e.bits(); i-- > 0;) {
      r = Fq2::sqr(r);
      if (e.bit(i)) r = Fq2::mul(r, a);
    }
    return r;
  }

  struct Consts {
    BigU final_exp;            // (q^6 + 1) / r
    Fq2 frob_x, frob_y;        // BN254: xi^((q-1)/3), xi^((q-1)/2)
  };
  static const Consts& consts() {
    static const Consts c = [] {
      Consts k;
      uint32_t ql[Fq::N], rl[Fr::N];
      for (int i = 0; i < Fq::N; i++) ql[i] = Fq::Params::mod(i);
      for (int i = 0; i < Fr::N; i++) rl[i] = Fr::Params::mod(i);
      const BigU q = BigU::from_u32(ql, Fq::N), r = BigU::from_u32(rl, Fr::N);
      const BigU q2 = BigU::mul(q, q), q6 = BigU::mul(BigU::mul(q2, q2), q2);
      BigU rem;
      k.final_exp = BigU::div(BigU::add(q6, BigU::small(1)), r, &rem);
      if (!rem.l.empty()) k.final_exp.l.clear();      // cannot happen: r | q^4 - q^2 + 1 | q^6 + 1
      const BigU qm1 = BigU::sub(q, BigU::small(1));
      const Fq2 xi = mul_xi(Fq2::one());
      k.frob_x = pow2(xi, BigU::div(qm1, BigU::small(3)));
      k.frob_y = pow2(xi, BigU::div(qm1, BigU::small(2)));
      return k;
    }();
    return c;
  }

  // the line through T and S (tangent when S == T) of the twist, evaluated at P = (xp, yp); T <- T + S
  static Fq12 line_and_add(Affine<Fq2>& T, const Affine<Fq2>& S, const Fq& xp, const Fq& yp) {
    Fq2 lam;
    if (T.x == S.x && T.y == S.y) lam = Fq2::mul(Fq2::mul3(Fq2::sqr(T.x)), Fq2::inv(Fq2::dbl(T.y)));
    else lam = Fq2::mul(Fq2::sub(S.y, T.y), Fq2::inv(Fq2::sub(S.x, T.x)));
    const Fq2 a = Fq2::sub(Fq2::mul(lam, T.x), T.y);         // lam x1 - y1
    const Fq2 b = Fq2::neg(mul_fq(lam, xp));                 // -lam xP
    const Fq2 ypp{yp, Fq::zero()};
    Fq12 l;
    if (BN) {        // D-type: yP + (-lam xP) w + (lam x1 - y1) v w
      l.c0 = Fq6{ypp, Fq2::zero(), Fq2::zero()};
      l.c1 = Fq6{b, a, Fq2::zero()};
    } else {         // M-type, scaled by w^3 (an element of F_q4, killed by the final exponentiation):
                     // (lam x1 - y1) + (-lam xP) v + yP v w
      l.c0 = Fq6{a, b, Fq2::zero()};
      l.c1 = Fq6{Fq2::zero(), ypp, Fq2::zero()};
    }
    const Fq2 x3 = Fq2::sub(Fq2::sub(Fq2::sqr(lam), T.x), S.x);
    const Fq2 y3 = Fq2::sub(Fq2::mul(lam, Fq2::sub(T.x, x3)), T.y);
    T = Affine<Fq2>{x3, y3};
    return l;
  }

  static Fq12 miller_loop(const Affine<Fq>& P, const Affine<Fq2>& Q) {
    if (P.is_inf() || Q.is_inf()) return Fq12::one();
    Affine<Fq2> T = Q;
    Fq12 f = Fq12::one();
    // loop count: |x| (BLS12-381) or 6x + 2 (BN254), most significant bit first, top bit skipped
    const uint64_t lo = BN ? 0x9D797039BE763BA8ull : 0xd201000000010000ull;
    const int top = BN ? 64 : 63;                    // BN254: 6x + 2 = 2^64 + lo
    for (int i = top - 1; i >= 0; i--) {
      f = Fq12::sqr(f);
      const Affine<Fq2> Tc = T;
      f = Fq12::mul(f, line_and_add(T, Tc, P.x, P.y));
      if ((lo >> i) & 1ull) f = Fq12::mul(f, line_and_add(T, Q, P.x, P.y));
    }
    if (BN) {
      const Consts& k = consts();
      auto frob = [&](const Affine<Fq2>& a) { return Affine<Fq2>{Fq2::mul(conj2(a.x), k.frob_x), Fq2::mul(conj2(a.y), k.frob_y)}; };
      const Affine<Fq2> Q1 = frob(Q);
      Affine<Fq2> Q2 = frob(Q1);
      Q2.y = Fq2::neg(Q2.y);
      f = Fq12::mul(f, line_and_add(T, Q1, P.x, P.y));
      f = Fq12::mul(f, line_and_add(T, Q2, P.x, P.y));
    } else {
      f = Fq12::conj(f);                             // x < 0
    }
    return f;
  }

  static Fq12 final_exponentiation(const Fq12& f) {
    const Fq12 f1 = Fq12::mul(Fq12::conj(f), Fq12::inv(f));          // f^(q^6 - 1)
    return Fq12::pow(f1, consts().final_exp);
  }

  // prod e(P_i, Q_i) == 1, Miller loops on `threads` host threads, one final exponentiation
  static bool product_is_one(const std::vector<Affine<Fq>>& Ps, const std::vector<Affine<Fq2>>& Qs, unsigned threads) {
    const size_t n = Ps.size();
    if (threads < 1) threads = 1;
    if (threads > n) threads = (unsigned)n;
    std::vector<Fq12> part(threads ? threads : 1, Fq12::one());
    (void)consts();                                   // build the constants before the threads start
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
      th.emplace_back([&, t] {
        Fq12 acc = Fq12::one();
        for (size_t i = t; i < n; i += threads) acc = Fq12::mul(acc, miller_loop(Ps[i], Qs[i]));
        part[t] = acc;
      });
    for (auto& x : th) x.join();
    Fq12 f = Fq12::one();
    for (const auto& p : part) f = Fq12::mul(f, p);
    if (consts().final_exp.l.empty()) return false;   // the exponent could not be formed: never accept
    return final_exponentiation(f) == Fq12::one();
  }
};

}  // namespace ark355
