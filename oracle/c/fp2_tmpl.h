/* TEST INFRASTRUCTURE (oracle): Fp2 = Fp[u]/(u^2+1) over the base field with prefix BF; new prefix F2.
 * Restates ark-ff QuadExtField arithmetic for the BLS12-381 / BN254 Fq2 (non-residue -1). */
#define C2_(a, b) a##b
#define C2(a, b) C2_(a, b)
#define B(name) C2(BF, name)
#define Q(name) C2(F2, name)

typedef struct {
  B(_t) c0, c1;
} Q(_t);
typedef B(_params) Q(_params);

static inline int Q(_is_zero)(const Q(_t) * a) { return B(_is_zero)(&a->c0) && B(_is_zero)(&a->c1); }
static inline int Q(_eq)(const Q(_t) * a, const Q(_t) * b) { return B(_eq)(&a->c0, &b->c0) && B(_eq)(&a->c1, &b->c1); }
static inline void Q(_add)(Q(_t) * r, const Q(_t) * a, const Q(_t) * b, const Q(_params) * P) {
  B(_add)(&r->c0, &a->c0, &b->c0, P);
  B(_add)(&r->c1, &a->c1, &b->c1, P);
}
static inline void Q(_sub)(Q(_t) * r, const Q(_t) * a, const Q(_t) * b, const Q(_params) * P) {
  B(_sub)(&r->c0, &a->c0, &b->c0, P);
  B(_sub)(&r->c1, &a->c1, &b->c1, P);
}
static inline void Q(_neg)(Q(_t) * r, const Q(_t) * a, const Q(_params) * P) {
  B(_neg)(&r->c0, &a->c0, P);
  B(_neg)(&r->c1, &a->c1, P);
}
static inline void Q(_dbl)(Q(_t) * r, const Q(_t) * a, const Q(_params) * P) { Q(_add)(r, a, a, P); }
static inline void Q(_mul)(Q(_t) * r, const Q(_t) * a, const Q(_t) * b, const Q(_params) * P) {
  B(_t) v0, v1, s, t;
  B(_mul)(&v0, &a->c0, &b->c0, P);
  B(_mul)(&v1, &a->c1, &b->c1, P);
  B(_add)(&s, &a->c0, &a->c1, P);
  B(_add)(&t, &b->c0, &b->c1, P);
  B(_mul)(&s, &s, &t, P);
  B(_sub)(&s, &s, &v0, P);
  B(_sub)(&r->c1, &s, &v1, P);
  B(_sub)(&r->c0, &v0, &v1, P);
}
static inline void Q(_sqr)(Q(_t) * r, const Q(_t) * a, const Q(_params) * P) { Q(_mul)(r, a, a, P); }
static inline void Q(_set_one)(Q(_t) * r, const Q(_params) * P) {
  B(_set_one)(&r->c0, P);
  B(_set_zero)(&r->c1);
}
static inline void Q(_set_zero)(Q(_t) * r) {
  B(_set_zero)(&r->c0);
  B(_set_zero)(&r->c1);
}
static void Q(_inv)(Q(_t) * r, const Q(_t) * a, const Q(_params) * P) {
  B(_t) n, t;
  B(_sqr)(&n, &a->c0, P);
  B(_sqr)(&t, &a->c1, P);
  B(_add)(&n, &n, &t, P);
  B(_inv)(&n, &n, P);
  B(_mul)(&r->c0, &a->c0, &n, P);
  B(_mul)(&t, &a->c1, &n, P);
  B(_neg)(&r->c1, &t, P);
}
#undef B
#undef Q
#undef C2
#undef C2_
