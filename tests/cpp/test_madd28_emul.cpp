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


// General additions on 28-bit slots (add28, add28_g2: the arithmetic of the tail kernels, tails28_impl.cuh) against the
// canonical 32-bit formulas: a pool of points that is summed pairwise at random (so that operands carry every value class
// a slot can hold: fresh mixed-addition results, sums of sums, negated openings), with P + P, P + (-P), P + O, O + P forced in.
template <class Curve>
static void run_add_g1(const char* name) {
  using Fq = typename Curve::Fq;
  using P = typename Fq::Params;
  using F28 = Fp28<P>;
  using Consts = typename Curve::Consts;
  Rng rng{0xadd1};
  Affine<Fq> g;
  for (int i = 0; i < Fq::N; i++) {
    g.x.l[i] = Consts::g1_gen_x(i);
    g.y.l[i] = Consts::g1_gen_y(i);
  }
  const size_t NP = 16;
  std::vector<XYZZ<Fq>> ref(NP);
  std::vector<Acc28<P>> acc(NP);
  std::vector<char> emp(NP);
  auto fresh = [&](size_t i) {
    uint32_t k[2] = {(uint32_t)rng.next() | 1u, (uint32_t)rng.next()};
    const Affine<Fq> a = xyzz_to_affine(xyzz_mul_scalar(XYZZ<Fq>::from_affine(g), k, 2));
    const bool ng = rng.next() & 1;
    Affine<Fq> q = a;
    if (ng) q.y = Fq::neg(q.y);
    ref[i] = XYZZ<Fq>::from_affine(q);
    acc[i].x = acc[i].y = acc[i].zz = acc[i].zzz = F28::zero();
    bool e = true;
    madd28<P>(acc[i], e, F28::from_fp(a.x), F28::from_fp(a.y), ng);       // opens the slot as the accumulation does
    emp[i] = e;
  };
  for (size_t i = 0; i < NP; i++) fresh(i);
  for (int step = 0; step < 30000; step++) {
    const size_t i = rng.next() % NP;
    size_t j = rng.next() % NP;
    const uint32_t kind = rng.next() % 40;
    Acc28<P> b = acc[j];
    bool be = emp[j] != 0;
    XYZZ<Fq> rb = ref[j];
    if (kind == 0) {                          // P + P through a slot round trip
      Slot28<P> s;
      slot28_put<P>(&s, acc[i], emp[i] != 0);
      be = slot28_get<P>(&s, b);
      rb = ref[i];
    } else if (kind == 1 && !emp[i]) {        // P + (-P)
      b = acc[i];
      b.y = F28::from_fp(Fq::neg(F28::to_fp(b.y)));       // canonical: stays inside the slot class y < 2p
      be = false;
      rb = XYZZ<Fq>::neg(ref[i]);
    } else if (kind == 2) {                   // + infinity
      be = true;
      rb = XYZZ<Fq>::inf();
    }
    bool ae = emp[i] != 0;
    add28<P>(acc[i], ae, b, be);
    emp[i] = ae;
    ref[i] = xyzz_add(ref[i], rb);
    CHECK(ae == ref[i].is_inf());
    if (!ae && (step % 7) == 0) {
      XYZZ<Fq> got{F28::to_fp(acc[i].x), F28::to_fp(acc[i].y), F28::to_fp(acc[i].zz), F28::to_fp(acc[i].zzz)};
      CHECK(same(xyzz_to_affine(got), xyzz_to_affine(ref[i])));
    }
    if (ae || (rng.next() % 64) == 0) fresh(i);
  }
  printf("%s G1 add28: ok\n", name);
}

template <class Curve>
static void run_add_g2(const char* name) {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using P = typename Fq::Params;
  using F28 = Fp28<P>;
  using Consts = typename Curve::Consts;
  Rng rng{0xadd2};
  Affine<Fq2> g;
  for (int i = 0; i < Fq::N; i++) {
    g.x.c0.l[i] = Consts::g2_gen_x0(i);
    g.x.c1.l[i] = Consts::g2_gen_x1(i);
    g.y.c0.l[i] = Consts::g2_gen_y0(i);
    g.y.c1.l[i] = Consts::g2_gen_y1(i);
  }
  // the script is generated up front: both lanes of the pair must see the same steps
  const size_t NP = 8, STEPS = 6000;
  struct Step {
    int op;            // 0: fresh(i) from point `a`, negated `ng`; 1: slot i += slot j (kind: 0 plain, 1 itself, 2 its negative, 3 infinity)
    size_t i, j;
    int kind;
    Affine<Fq2> a;
    bool ng;
  };
  std::vector<Step> script;
  std::vector<XYZZ<Fq2>> ref(NP, XYZZ<Fq2>::inf());
  std::vector<XYZZ<Fq2>> ref_after;
  auto gen_fresh = [&](size_t i) {
    uint32_t k[2] = {(uint32_t)rng.next() | 1u, (uint32_t)rng.next()};
    Step s{0, i, 0, 0, xyzz_to_affine(xyzz_mul_scalar(XYZZ<Fq2>::from_affine(g), k, 2)), (bool)(rng.next() & 1)};
    Affine<Fq2> q = s.a;
    if (s.ng) q.y = Fq2::neg(q.y);
    ref[i] = XYZZ<Fq2>::from_affine(q);
    script.push_back(s);
    ref_after.push_back(ref[i]);
  };
  for (size_t i = 0; i < NP; i++) gen_fresh(i);
  for (size_t t = 0; t < STEPS; t++) {
    Step s{1, (size_t)(rng.next() % NP), (size_t)(rng.next() % NP), 0, Affine<Fq2>::inf(), false};
    const uint32_t kind = rng.next() % 30;
    s.kind = kind == 0 ? 1 : (kind == 1 && !ref[s.i].is_inf()) ? 2 : kind == 2 ? 3 : 0;
    const XYZZ<Fq2> rb = s.kind == 0 ? ref[s.j] : s.kind == 1 ? ref[s.i] : s.kind == 2 ? XYZZ<Fq2>::neg(ref[s.i]) : XYZZ<Fq2>::inf();
    ref[s.i] = xyzz_add(ref[s.i], rb);
    script.push_back(s);
    ref_after.push_back(ref[s.i]);
    if (ref[s.i].is_inf() || (rng.next() % 48) == 0) gen_fresh(s.i);
  }
  std::vector<XYZZ<Fq2>> got_after(script.size());
  std::vector<char> got_empty(script.size());
  emu::launch(dim3(1), dim3(2), 0, [&]() {
    const uint32_t par = threadIdx.x & 1u;
    std::vector<Acc28<P>> acc(NP);
    std::vector<char> emp(NP, 1);
    for (size_t t = 0; t < script.size(); t++) {
      const Step& s = script[t];
      if (s.op == 0) {
        acc[s.i].x = acc[s.i].y = acc[s.i].zz = acc[s.i].zzz = F28::zero();
        bool e = true;
        madd28_g2<P>(acc[s.i], e, F28::from_fp(par ? s.a.x.c1 : s.a.x.c0), F28::from_fp(par ? s.a.y.c1 : s.a.y.c0), s.ng);
        emp[s.i] = e;
      } else {
        Acc28<P> b = acc[s.j];
        bool be = emp[s.j] != 0;
        if (s.kind == 1) {
          Slot28G2<P> slot;                    // (each lane writes and reads its own half)
          slot28_put<P>(&slot.half[par], acc[s.i], emp[s.i] != 0);
          const bool z = slot28_get<P>(&slot.half[par], b);
          be = Pair28<P>::both(z);
        } else if (s.kind == 2) {
          b = acc[s.i];
          b.y = F28::from_fp(Fq::neg(F28::to_fp(b.y)));       // canonical: stays inside the slot class y < 2p
          be = false;
        } else if (s.kind == 3) {
          be = true;
        }
        bool ae = emp[s.i] != 0;
        add28_g2<P>(acc[s.i], ae, b, be);
        emp[s.i] = ae;
      }
      if (par == 0) got_empty[t] = emp[s.i];
      if (!emp[s.i] && (acc[s.i].x.l[F28::N - 1] > 8u * F28::template kp<1>(F28::N - 1) || acc[s.i].y.l[F28::N - 1] > 3u * F28::template kp<1>(F28::N - 1))) {
        fprintf(stderr, "slot class violated after step %zu: op %d kind %d (x top %08x, y top %08x)\n", t, s.op, s.kind,
                acc[s.i].x.l[F28::N - 1], acc[s.i].y.l[F28::N - 1]);
        exit(1);
      }
      if (!emp[s.i]) {
        Fq* d = reinterpret_cast<Fq*>(&got_after[t]);
        d[0 + par] = F28::to_fp(acc[s.i].x);
        d[2 + par] = F28::to_fp(acc[s.i].y);
        d[4 + par] = F28::to_fp(acc[s.i].zz);
        d[6 + par] = F28::to_fp(acc[s.i].zzz);
      }
    }
  });
  for (size_t t = 0; t < script.size(); t++) {
    CHECK((got_empty[t] != 0) == ref_after[t].is_inf());
    if (!ref_after[t].is_inf() && ((t % 5) == 0 || t + 1 == script.size()))
      CHECK(same(xyzz_to_affine(got_after[t]), xyzz_to_affine(ref_after[t])));
  }
  printf("%s G2 add28 (lane pair): ok\n", name);
}

int main() {
  run_kara<BlsFq28>("bls12_381");
  run_kara<BnFq28>("bn254");
  run_g1<BlsCurve>("bls12_381");
  run_g1<BnCurve>("bn254");
  run_g2<BlsCurve>("bls12_381");
  run_g2<BnCurve>("bn254");
  run_add_g1<BlsCurve>("bls12_381");
  run_add_g1<BnCurve>("bn254");
  run_add_g2<BlsCurve>("bls12_381");
  run_add_g2<BnCurve>("bn254");
  printf("madd28 stress: all chains agree with the 32-bit formulas\n");
  return 0;
}
