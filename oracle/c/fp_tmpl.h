/* TEST INFRASTRUCTURE (oracle): Montgomery prime field on NL 64-bit limbs, instantiated by inclusion.
 *
 * Restates the published algorithm of ark-ff `MontBackend` (un-vendored crate
 * ark-ff/src/fields/models/fp/montgomery_backend.rs): CIOS multiplication on u64 limbs with
 * 128-bit products, R = 2^(64*NL).  Independent of the product's 32-bit-limb device code.
 *
 * Before including define:  NL (limbs), FP (name prefix).
 */
#include <stdint.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(FP, name)

typedef struct {
  uint64_t l[NL];
} FN(_t);

typedef struct {
  uint64_t mod[NL];
  uint64_t one[NL]; /* R mod p */
  uint64_t r2[NL];  /* R^2 mod p */
  uint64_t inv;     /* -p^-1 mod 2^64 */
} FN(_params);

static inline int FN(_is_zero)(const FN(_t) * a) {
  uint64_t acc = 0;
  for (int i = 0; i < NL; i++) acc |= a->l[i];
  return acc == 0;
}
static inline int FN(_eq)(const FN(_t) * a, const FN(_t) * b) {
  uint64_t acc = 0;
  for (int i = 0; i < NL; i++) acc |= a->l[i] ^ b->l[i];
  return acc == 0;
}
static inline int FN(_geq_mod)(const uint64_t* a, const FN(_params) * P) {
  for (int i = NL - 1; i >= 0; i--) {
    if (a[i] > P->mod[i]) return 1;
    if (a[i] < P->mod[i]) return 0;
  }
  return 1;
}
static inline void FN(_sub_mod)(uint64_t* a, const FN(_params) * P) {
  unsigned __int128 br = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 t = (unsigned __int128)a[i] - P->mod[i] - (uint64_t)br;
    a[i] = (uint64_t)t;
    br = (t >> 64) & 1;
  }
}
static inline void FN(_add)(FN(_t) * r, const FN(_t) * a, const FN(_t) * b, const FN(_params) * P) {
  unsigned __int128 c = 0;
  uint64_t t[NL];
  for (int i = 0; i < NL; i++) {
    c += (unsigned __int128)a->l[i] + b->l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || FN(_geq_mod)(t, P)) FN(_sub_mod)(t, P);
  memcpy(r->l, t, sizeof(t));
}
static inline void FN(_sub)(FN(_t) * r, const FN(_t) * a, const FN(_t) * b, const FN(_params) * P) {
  unsigned __int128 br = 0;
  uint64_t t[NL];
  for (int i = 0; i < NL; i++) {
    unsigned __int128 x = (unsigned __int128)a->l[i] - b->l[i] - (uint64_t)br;
    t[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  if (br) {
    unsigned __int128 c = 0;
    for (int i = 0; i < NL; i++) {
      c += (unsigned __int128)t[i] + P->mod[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r->l, t, sizeof(t));
}
static inline void FN(_neg)(FN(_t) * r, const FN(_t) * a, const FN(_params) * P) {
  if (FN(_is_zero)(a)) {
    *r = *a;
    return;
  }
  unsigned __int128 br = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 x = (unsigned __int128)P->mod[i] - a->l[i] - (uint64_t)br;
    r->l[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
}
static inline void FN(_dbl)(FN(_t) * r, const FN(_t) * a, const FN(_params) * P) { FN(_add)(r, a, a, P); }

static inline void FN(_mul)(FN(_t) * r, const FN(_t) * a, const FN(_t) * b, const FN(_params) * P) {
  uint64_t t[NL + 2];
  memset(t, 0, sizeof(t));
  for (int i = 0; i < NL; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < NL; j++) {
      c += (unsigned __int128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[NL];
    t[NL] = (uint64_t)c;
    t[NL + 1] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * P->inv;
    c = (unsigned __int128)m * P->mod[0] + t[0];
    c >>= 64;
    for (int j = 1; j < NL; j++) {
      c += (unsigned __int128)m * P->mod[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[NL];
    t[NL - 1] = (uint64_t)c;
    t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
  }
  if (t[NL] || FN(_geq_mod)(t, P)) FN(_sub_mod)(t, P);
  memcpy(r->l, t, NL * sizeof(uint64_t));
}
static inline void FN(_sqr)(FN(_t) * r, const FN(_t) * a, const FN(_params) * P) { FN(_mul)(r, a, a, P); }

static inline void FN(_set_one)(FN(_t) * r, const FN(_params) * P) { memcpy(r->l, P->one, sizeof(r->l)); }
static inline void FN(_set_zero)(FN(_t) * r) { memset(r->l, 0, sizeof(r->l)); }

/* a^e, e little-endian u64 limbs */
static void FN(_pow)(FN(_t) * r, const FN(_t) * a, const uint64_t* e, int elimbs, const FN(_params) * P) {
  FN(_t) acc;
  FN(_set_one)(&acc, P);
  for (int i = elimbs - 1; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      FN(_sqr)(&acc, &acc, P);
      if ((e[i] >> b) & 1) FN(_mul)(&acc, &acc, a, P);
    }
  *r = acc;
}
static void FN(_inv)(FN(_t) * r, const FN(_t) * a, const FN(_params) * P) {
  uint64_t e[NL];
  memcpy(e, P->mod, sizeof(e));
  /* p - 2 (p is odd and > 2, no borrow beyond limb 0 unless limb0 < 2) */
  unsigned __int128 br = 2;
  for (int i = 0; i < NL && br; i++) {
    unsigned __int128 x = (unsigned __int128)e[i] - (uint64_t)br;
    e[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  FN(_pow)(r, a, e, NL, P);
}
static inline void FN(_from_mont)(FN(_t) * r, const FN(_t) * a, const FN(_params) * P) {
  FN(_t) o;
  FN(_set_zero)(&o);
  o.l[0] = 1;
  FN(_mul)(r, a, &o, P);
}
static inline void FN(_to_mont)(FN(_t) * r, const FN(_t) * a, const FN(_params) * P) {
  FN(_t) r2;
  memcpy(r2.l, P->r2, sizeof(r2.l));
  FN(_mul)(r, a, &r2, P);
}

#undef FN
#undef CAT
#undef CAT_
