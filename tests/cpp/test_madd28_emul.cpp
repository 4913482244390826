// TEST INFRASTRUCTURE (CPU tier): long randomised accumulation chains through the radix-2^28 mixed additions
// (madd28: G1, madd28_g2: lane-split G2 on an emulated lane pair) against the canonical 32-bit formulas, with the
// exceptional cases forced in (P + P, P + (-P), restart after infinity).  Built with -DARK_EMUL, so every column
// accumulator and every lazy limb operation is checked for overflow / wrap-around on the way (field28.cuh).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "msm_impl.cuh"

using namespace ark355;

#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

struct Rng {
  uint64_t s;
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
};

template <class F>
static bool same(const Affine<F>& a, const Affine<F>& b) { return a.x == b.x && a.y == b.y; }

template <class Curve>
static void run_g1(const char* name) {
  using Fq = typename Curve::Fq;
  using P = typename Fq::Params;
  using F28 = Fp28<P>;
  using Consts = typename Curve::Consts;
  Rng rng{0x355};
  Affine<Fq> g;
  for (int i = 0; i < Fq::N; i++) {
    g.x.l[i] = Consts::g1_gen_x(i);
    g.y.l[i] = Consts::g1_gen_y(i);
  }
  const size_t NP = 24;
  std::vector<Affine<Fq>> pts;
  for (size_t i = 0; i < NP; i++) {
    uint32_t k[2] = {(uint32_t)rng.next() | 1u, (uint32_t)rng.next()};
    pts.push_back(xyzz_to_affine(xyzz_mul_scalar(XYZZ<Fq>::from_affine(g), k, 2)));
  }
  for (int trial = 0; trial < 12; trial++) {
    XYZZ<Fq> ref = XYZZ<Fq>::inf();
    Acc28<P> acc;
    acc.x = acc.y = acc.zz = acc.zzz = F28::zero();
    bool empty = true;
    const size_t LEN = 4000;
    for (size_t s = 0; s < LEN; s++) {
      uint32_t i = rng.next() % NP;
      bool ng = rng.next() & 1;
      const uint32_t kind = rng.next() % 50;
      Affine<Fq> p = pts[i];
      if (kind == 0 && !ref.is_inf()) {               // P == acc: add the current sum itself (forces the doubling path)
        p = xyzz_to_affine(ref);
        ng = false;
      } else if (kind == 1 && !ref.is_inf()) {        // P == -acc: back to infinity
        p = xyzz_to_affine(ref);
        ng = true;
      }
      F28 px = F28::from_fp(p.x), py = F28::from_fp(p.y);
      Affine<Fq> q = p;
      if (ng) q.y = Fq::neg(q.y);
      xyzz_madd(ref, q);
      madd28<P>(acc, empty, px, py, ng);
      CHECK(empty == ref.is_inf());
      if ((s % 97) == 0 || s + 1 == LEN) {
        if (!empty) {
          XYZZ<Fq> got{F28::to_fp(acc.x), F28::to_fp(acc.y), F28::to_fp(acc.zz), F28::to_fp(acc.zzz)};
          CHECK(same(xyzz_to_affine(got), xyzz_to_affine(ref)));
        }
      }
    }
  }
  printf("%s G1: ok\n", name);
}

template <class Curve>
static void run_g2(const char* name) {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using P = typename Fq::Params;
  using F28 = Fp28<P>;
  using Consts = typename Curve::Consts;
  Rng rng{0x77};
  Affine<Fq2> g;
  for (int i = 0; i < Fq::N; i++) {
    g.x.c0.l[i] = Consts::g2_gen_x0(i);
    g.x.c1.l[i] = Consts::g2_gen_x1(i);
    g.y.c0.l[i] = Consts::g2_gen_y0(i);
    g.y.c1.l[i] = Consts::g2_gen_y1(i);
  }
  const size_t NP = 12;
  std::vector<Affine<Fq2>> pts;
  for (size_t i = 0; i < NP; i++) {
    uint32_t k[2] = {(uint32_t)rng.next() | 1u, (uint32_t)rng.next()};
    pts.push_back(xyzz_to_affine(xyzz_mul_scalar(XYZZ<Fq2>::from_affine(g), k, 2)));
  }
  // the chain is generated up front (both lanes must see the same steps)
  const size_t LEN = 6000;
  std::vector<Affine<Fq2>> chain_p(LEN);
  std::vector<char> chain_neg(LEN);
  std::vector<XYZZ<Fq2>> ref_after(LEN);
  XYZZ<Fq2> ref = XYZZ<Fq2>::inf();
  for (size_t s = 0; s < LEN; s++) {
    uint32_t i = rng.next() % NP;
    bool ng = rng.next() & 1;
    const uint32_t kind = rng.next() % 40;
    Affine<Fq2> p = pts[i];
    if (kind == 0 && !ref.is_inf()) {
      p = xyzz_to_affine(ref);
      ng = false;
    } else if (kind == 1 && !ref.is_inf()) {
      p = xyzz_to_affine(ref);
      ng = true;
    }
    chain_p[s] = p;
    chain_neg[s] = ng;
    Affine<Fq2> q = p;
    if (ng) q.y = Fq2::neg(q.y);
    xyzz_madd_ni(ref, q);
    ref_after[s] = ref;
  }
  // two emulated lanes: lane parity = Fq2 component
  std::vector<XYZZ<Fq2>> got_after(LEN);
  std::vector<char> got_empty(LEN);
  emu::launch(dim3(1), dim3(2), 0, [&]() {
    const uint32_t par = threadIdx.x & 1u;
    Acc28<P> acc;
    acc.x = acc.y = acc.zz = acc.zzz = F28::zero();
    bool empty = true;
    for (size_t s = 0; s < LEN; s++) {
      const Affine<Fq2>& p = chain_p[s];
      const F28 px = F28::from_fp(par ? p.x.c1 : p.x.c0), py = F28::from_fp(par ? p.y.c1 : p.y.c0);
      madd28_g2<P>(acc, empty, px, py, chain_neg[s] != 0);
      if (par == 0) got_empty[s] = empty;
      Fq* d = reinterpret_cast<Fq*>(&got_after[s]);
      if (!empty) {
        d[0 + par] = F28::to_fp(acc.x);
        d[2 + par] = F28::to_fp(acc.y);
        d[4 + par] = F28::to_fp(acc.zz);
        d[6 + par] = F28::to_fp(acc.zzz);
      }
    }
  });
  for (size_t s = 0; s < LEN; s++) {
    CHECK((got_empty[s] != 0) == ref_after[s].is_inf());
    if (!ref_after[s].is_inf() && ((s % 13) == 0 || s + 1 == LEN))
      CHECK(same(xyzz_to_affine(got_after[s]), xyzz_to_affine(ref_after[s])));
  }
  printf("%s G2 (lane pair): ok\n", name);
}

// One Karatsuba level (mulsum_kara, field28.cuh) against the schoolbook pass it replaces: single, dual and quad product
// passes, operand limbs at the widest classes each pass admits (x + y <= 60 / 59 / 58), all-ones and half-zero patterns
// (the signed middle term at its extremes).  Limb for limb, not just modulo p; every column is checked against its
// 128-bit shadow on the way (ARK_EMUL).
template <class F>
static void run_kara(const char* name) {
  Rng rng{0x5a5a};
  for (int t = 0; t < 60000; t++) {
    F a[4], b[4];
    const int bits = 28 + (t % 3);
    for (int j = 0; j < 4; j++)
      for (int i = 0; i < F::N; i++) {
        a[j].l[i] = (uint32_t)rng.next() & ((1u << bits) - 1);
        b[j].l[i] = (uint32_t)rng.next() & F::MASK;
      }
    if (t % 7 == 0)
      for (int i = 0; i < F::N; i++) {
        a[0].l[i] = (1u << bits) - 1;
        b[0].l[i] = F::MASK;
      }
    if (t % 11 == 0)
      for (int i = 0; i < F::N; i++) {
        a[0].l[i] = (i < F::H) ? (1u << bits) - 1 : 0;
        b[0].l[i] = (i < F::H) ? 0 : F::MASK;
      }
    auto same_limbs = [](const F& x, const F& y) {
      for (int i = 0; i < F::N; i++)
        if (x.l[i] != y.l[i]) return false;
      return true;
    };
    {
      const F* const x[1] = {&a[0]};
      const F* const y[1] = {&b[0]};
      CHECK(same_limbs(F::template mulsum_school<1>(x, y), F::template mulsum_kara<1>(x, y)));
    }
    if (bits <= 29) {
      const F* const x[2] = {&a[0], &a[1]};
      const F* const y[2] = {&b[0], &b[1]};
      CHECK(same_limbs(F::template mulsum_school<2>(x, y), F::template mulsum_kara<2>(x, y)));
    }
    if (bits <= 28) {
      const F* const x[4] = {&a[0], &a[1], &a[2], &a[3]};
      const F* const y[4] = {&b[0], &b[1], &b[2], &b[3]};
      CHECK(same_limbs(F::template mulsum_school<4>(x, y), F::template mulsum_kara<4>(x, y)));
    }
  }
  printf("%s Karatsuba == schoolbook: ok\n", name);
}

int main() {
  run_kara<BlsFq28>("bls12_381");
  run_kara<BnFq28>("bn254");
  run_g1<BlsCurve>("bls12_381");
  run_g1<BnCurve>("bn254");
  run_g2<BlsCurve>("bls12_381");
  run_g2<BnCurve>("bn254");
  printf("madd28 stress: all chains agree with the 32-bit formulas\n");
  return 0;
}
