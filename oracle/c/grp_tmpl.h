/* TEST INFRASTRUCTURE (oracle): short-Weierstrass a=0 group in Jacobian coordinates + arkworks-style
 * Pippenger, instantiated by inclusion.  Define FEP (field prefix, e.g. f6 or f6x2) and GP (group prefix).
 *
 * Restates the published algorithms of ark-ec `short_weierstrass::Projective` (Jacobian add / mixed add /
 * double, EFD add-2007-bl / madd-2007-bl / dbl-2009-l) and `VariableBaseMSM::msm_bigint_wnaf`
 * (ark-ec/src/scalar_mul/variable_base/mod.rs): window c = 3 if n < 32 else ln(n) + 2 with
 * ln(n) ~ log2(n) * 69 / 100, signed digits, 2^(c-1) buckets per window, windows processed in
 * parallel (rayon there, OpenMP here), running-sum bucket reduction, Horner combination of windows.
 */
#define G2_(a, b) a##b
#define G2(a, b) G2_(a, b)
#define F(name) G2(FEP, name)
#define G(name) G2(GP, name)

typedef struct {
  F(_t) x, y;
} G(_aff);   /* infinity: x == 0 && y == 0 */
typedef struct {
  F(_t) x, y, z;
} G(_jac);   /* infinity: z == 0 */

static inline int G(_aff_is_inf)(const G(_aff) * p) { return F(_is_zero)(&p->x) && F(_is_zero)(&p->y); }
static inline void G(_set_inf)(G(_jac) * p, const F(_params) * P) {
  F(_set_one)(&p->x, P);
  F(_set_one)(&p->y, P);
  F(_set_zero)(&p->z);
}
static inline int G(_is_inf)(const G(_jac) * p) { return F(_is_zero)(&p->z); }

static void G(_double)(G(_jac) * r, const G(_jac) * p, const F(_params) * P) {
  if (G(_is_inf)(p) || F(_is_zero)(&p->y)) {
    G(_set_inf)(r, P);
    return;
  }
  F(_t) a, b, c, d, e, f, t;
  F(_sqr)(&a, &p->x, P);
  F(_sqr)(&b, &p->y, P);
  F(_sqr)(&c, &b, P);
  F(_add)(&t, &p->x, &b, P);
  F(_sqr)(&t, &t, P);
  F(_sub)(&t, &t, &a, P);
  F(_sub)(&t, &t, &c, P);
  F(_dbl)(&d, &t, P);
  F(_dbl)(&e, &a, P);
  F(_add)(&e, &e, &a, P);
  F(_sqr)(&f, &e, P);
  F(_t) z3;
  F(_mul)(&z3, &p->y, &p->z, P);
  F(_dbl)(&z3, &z3, P);
  F(_dbl)(&t, &d, P);
  F(_sub)(&r->x, &f, &t, P);
  F(_sub)(&t, &d, &r->x, P);
  F(_mul)(&t, &e, &t, P);
  F(_dbl)(&c, &c, P);
  F(_dbl)(&c, &c, P);
  F(_dbl)(&c, &c, P);
  F(_sub)(&r->y, &t, &c, P);
  r->z = z3;
}

static void G(_add)(G(_jac) * r, const G(_jac) * p, const G(_jac) * q, const F(_params) * P) {
  if (G(_is_inf)(p)) {
    *r = *q;
    return;
  }
  if (G(_is_inf)(q)) {
    *r = *p;
    return;
  }
  F(_t) z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t;
  F(_sqr)(&z1z1, &p->z, P);
  F(_sqr)(&z2z2, &q->z, P);
  F(_mul)(&u1, &p->x, &z2z2, P);
  F(_mul)(&u2, &q->x, &z1z1, P);
  F(_mul)(&s1, &q->z, &z2z2, P);
  F(_mul)(&s1, &s1, &p->y, P);
  F(_mul)(&s2, &p->z, &z1z1, P);
  F(_mul)(&s2, &s2, &q->y, P);
  if (F(_eq)(&u1, &u2)) {
    if (F(_eq)(&s1, &s2)) {
      G(_double)(r, p, P);
    } else {
      G(_set_inf)(r, P);
    }
    return;
  }
  F(_sub)(&h, &u2, &u1, P);
  F(_sub)(&rr, &s2, &s1, P);
  F(_sqr)(&hh, &h, P);
  F(_mul)(&hhh, &h, &hh, P);
  F(_mul)(&v, &u1, &hh, P);
  F(_t) x3, y3, z3;
  F(_sqr)(&x3, &rr, P);
  F(_sub)(&x3, &x3, &hhh, P);
  F(_dbl)(&t, &v, P);
  F(_sub)(&x3, &x3, &t, P);
  F(_sub)(&t, &v, &x3, P);
  F(_mul)(&y3, &rr, &t, P);
  F(_mul)(&t, &s1, &hhh, P);
  F(_sub)(&y3, &y3, &t, P);
  F(_mul)(&z3, &p->z, &q->z, P);
  F(_mul)(&z3, &z3, &h, P);
  r->x = x3;
  r->y = y3;
  r->z = z3;
}

static void G(_madd)(G(_jac) * r, const G(_jac) * p, const G(_aff) * q, const F(_params) * P) {
  if (G(_aff_is_inf)(q)) {
    *r = *p;
    return;
  }
  if (G(_is_inf)(p)) {
    r->x = q->x;
    r->y = q->y;
    F(_set_one)(&r->z, P);
    return;
  }
  F(_t) z1z1, u2, s2, h, rr, hh, hhh, v, t;
  F(_sqr)(&z1z1, &p->z, P);
  F(_mul)(&u2, &q->x, &z1z1, P);
  F(_mul)(&s2, &p->z, &z1z1, P);
  F(_mul)(&s2, &s2, &q->y, P);
  if (F(_eq)(&p->x, &u2)) {
    if (F(_eq)(&p->y, &s2)) {
      G(_double)(r, p, P);
    } else {
      G(_set_inf)(r, P);
    }
    return;
  }
  F(_sub)(&h, &u2, &p->x, P);
  F(_sub)(&rr, &s2, &p->y, P);
  F(_sqr)(&hh, &h, P);
  F(_mul)(&hhh, &h, &hh, P);
  F(_mul)(&v, &p->x, &hh, P);
  F(_t) x3, y3, z3;
  F(_sqr)(&x3, &rr, P);
  F(_sub)(&x3, &x3, &hhh, P);
  F(_dbl)(&t, &v, P);
  F(_sub)(&x3, &x3, &t, P);
  F(_sub)(&t, &v, &x3, P);
  F(_mul)(&y3, &rr, &t, P);
  F(_mul)(&t, &p->y, &hhh, P);
  F(_sub)(&y3, &y3, &t, P);
  F(_mul)(&z3, &p->z, &h, P);
  r->x = x3;
  r->y = y3;
  r->z = z3;
}

static void G(_to_affine)(G(_aff) * r, const G(_jac) * p, const F(_params) * P) {
  if (G(_is_inf)(p)) {
    F(_set_zero)(&r->x);
    F(_set_zero)(&r->y);
    return;
  }
  F(_t) zi, zi2, zi3;
  F(_inv)(&zi, &p->z, P);
  F(_sqr)(&zi2, &zi, P);
  F(_mul)(&zi3, &zi2, &zi, P);
  F(_mul)(&r->x, &p->x, &zi2, P);
  F(_mul)(&r->y, &p->y, &zi3, P);
}

/* k * p, k = 4 canonical u64 limbs (256-bit), plain double-and-add */
static void G(_mul_scalar)(G(_jac) * r, const G(_jac) * p, const uint64_t k[4], const F(_params) * P) {
  G(_jac) acc;
  G(_set_inf)(&acc, P);
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      G(_double)(&acc, &acc, P);
      if ((k[i] >> b) & 1) G(_add)(&acc, &acc, p, P);
    }
  *r = acc;
}

/* arkworks-style Pippenger over canonical 256-bit scalars */
static void G(_msm)(G(_jac) * out, const G(_aff) * bases, const uint64_t* scalars /* n x 4 */, size_t n,
                    int scalar_bits, const F(_params) * P) {
  G(_set_inf)(out, P);
  if (n == 0) return;
  int c;
  if (n < 32) c = 3;
  else {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    /* ark_std::log2(n) is ceil(log2 n) */
    if (((size_t)1 << lg) < n) lg++;
    c = lg * 69 / 100 + 2;
  }
  const int nwin = (scalar_bits + c - 1) / c + 1;   /* make_digits: one extra window for the carry */
  const size_t nb = (size_t)1 << (c - 1);
  /* signed digits, window-major */
  int32_t* digits = (int32_t*)malloc(sizeof(int32_t) * n * (size_t)nwin);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    const uint64_t* k = scalars + 4 * i;
    int64_t carry = 0;
    for (int w = 0; w < nwin; w++) {
      const int bit = w * c;
      uint64_t d = 0;
      if (bit < 256) {
        const int limb = bit >> 6, off = bit & 63;
        d = k[limb] >> off;
        if (off + c > 64 && limb + 1 < 4) d |= k[limb + 1] << (64 - off);
        d &= (((uint64_t)1 << c) - 1);
      }
      int64_t v = (int64_t)d + carry;
      carry = (v + ((int64_t)1 << (c - 1))) >> c;       /* arkworks: carry = (digit + radix/2) >> w */
      v -= carry << c;
      digits[(size_t)w * n + i] = (int32_t)v;
    }
  }
  /* Tasks = (window, chunk of the terms).  arkworks parallelises over windows (rayon); with ~18 windows that leaves
   * most of a 128-thread host idle, while its Groth16 prover gets further parallelism from running the MSMs of a
   * proof side by side.  Splitting every window's terms into chunks reaches the same occupancy inside one MSM; the
   * per-window sum is the same group element whatever the partition. */
  int nthreads = omp_get_max_threads();
  size_t nchunks = ((size_t)2 * (size_t)nthreads + (size_t)nwin - 1) / (size_t)nwin;
  if (nchunks * 4 * nb > n) nchunks = n / (4 * nb);      /* keep the bucket reduction of a task below ~half its work */
  if (nchunks < 1) nchunks = 1;
  const size_t ntasks = (size_t)nwin * nchunks;
  G(_jac)* tsum = (G(_jac)*)malloc(sizeof(G(_jac)) * ntasks);
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t task = 0; task < ntasks; task++) {
    const int w = (int)(task / nchunks);
    const size_t ch = task % nchunks;
    const size_t i0 = n * ch / nchunks, i1 = n * (ch + 1) / nchunks;
    G(_jac)* buckets = (G(_jac)*)malloc(sizeof(G(_jac)) * nb);
    for (size_t b = 0; b < nb; b++) G(_set_inf)(&buckets[b], P);
    const int32_t* dg = digits + (size_t)w * n;
    for (size_t i = i0; i < i1; i++) {
      int32_t d = dg[i];
      if (d > 0) {
        G(_madd)(&buckets[d - 1], &buckets[d - 1], &bases[i], P);
      } else if (d < 0) {
        G(_aff) neg = bases[i];
        F(_neg)(&neg.y, &neg.y, P);
        G(_madd)(&buckets[-d - 1], &buckets[-d - 1], &neg, P);
      }
    }
    G(_jac) running, res;
    G(_set_inf)(&running, P);
    G(_set_inf)(&res, P);
    for (size_t b = nb; b-- > 0;) {
      G(_add)(&running, &running, &buckets[b], P);
      G(_add)(&res, &res, &running, P);
    }
    tsum[task] = res;
    free(buckets);
  }
  G(_jac)* wsum = (G(_jac)*)malloc(sizeof(G(_jac)) * nwin);
  for (int w = 0; w < nwin; w++) {
    G(_set_inf)(&wsum[w], P);
    for (size_t ch = 0; ch < nchunks; ch++) G(_add)(&wsum[w], &wsum[w], &tsum[(size_t)w * nchunks + ch], P);
  }
  free(tsum);
  /* lowest + fold the rest from the top (Horner, c doublings per window) */
  G(_jac) total;
  G(_set_inf)(&total, P);
  for (int w = nwin - 1; w >= 1; w--) {
    G(_add)(&total, &total, &wsum[w], P);
    for (int j = 0; j < c; j++) G(_double)(&total, &total, P);
  }
  G(_add)(out, &total, &wsum[0], P);
  free(wsum);
  free(digits);
}

/* out[i] = k_i * base via a fixed-base window table (setup; ark-ec FixedBase::msm: windowed table, one mixed addition per
 * window, batch normalisation).  8-bit windows for short vectors, 16-bit windows (16 additions per scalar, a table of 2^20
 * points built in parallel) from 2^16 scalars on. */
static void G(_normalize_chunk)(G(_aff) * out, const G(_jac) * in, size_t cnt, F(_t) * pref, const F(_params) * P) {
  F(_t) run, inv;
  F(_set_one)(&run, P);
  for (size_t k = 0; k < cnt; k++) {
    pref[k] = run;
    if (!G(_is_inf)(&in[k])) F(_mul)(&run, &run, &in[k].z, P);
  }
  F(_inv)(&inv, &run, P);
  for (size_t k = cnt; k-- > 0;) {
    if (G(_is_inf)(&in[k])) {
      F(_set_zero)(&out[k].x);
      F(_set_zero)(&out[k].y);
      continue;
    }
    F(_t) zi, zi2, zi3;
    F(_mul)(&zi, &inv, &pref[k], P);               /* 1 / z_k */
    F(_mul)(&inv, &inv, &in[k].z, P);
    F(_sqr)(&zi2, &zi, P);
    F(_mul)(&zi3, &zi2, &zi, P);
    F(_mul)(&out[k].x, &in[k].x, &zi2, P);
    F(_mul)(&out[k].y, &in[k].y, &zi3, P);
  }
}

static void G(_fixed_base)(G(_aff) * out, const G(_aff) * base, const uint64_t* scalars, size_t n,
                           const F(_params) * P) {
  const int WB = n >= ((size_t)1 << 16) ? 16 : 8;
  const int NW = 256 / WB;
  const size_t TW = (size_t)1 << WB;                  /* entries per window */
  enum { CH = 512 };
  G(_aff)* table = (G(_aff)*)malloc(sizeof(G(_aff)) * NW * TW);
  G(_jac)* wbase = (G(_jac)*)malloc(sizeof(G(_jac)) * NW);
  {
    G(_jac) cur;
    cur.x = base->x;
    cur.y = base->y;
    F(_set_one)(&cur.z, P);
    if (G(_aff_is_inf)(base)) G(_set_inf)(&cur, P);
    for (int w = 0; w < NW; w++) {
      wbase[w] = cur;
      for (int j = 0; j < WB; j++) G(_double)(&cur, &cur, P);
    }
  }
  /* table[w][d] = d * 2^(WB w) * base, normalised chunk by chunk (one inversion per CH entries) */
  const size_t tchunks = (TW + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int w = 0; w < NW; w++) {
    for (size_t tc = 0; tc < tchunks; tc++) {
      const size_t d0 = tc * CH, d1 = (d0 + CH < TW) ? d0 + CH : TW;
      G(_jac) tj[CH];
      F(_t) pref[CH];
      G(_jac) acc;
      uint64_t k[4] = {(uint64_t)d0, 0, 0, 0};
      G(_mul_scalar)(&acc, &wbase[w], k, P);           /* d0 * wbase (d0 = 0: infinity) */
      for (size_t d = d0; d < d1; d++) {
        tj[d - d0] = acc;
        G(_add)(&acc, &acc, &wbase[w], P);
      }
      G(_normalize_chunk)(&table[(size_t)w * TW + d0], tj, d1 - d0, pref, P);
    }
  }
  free(wbase);
  const size_t nchunks = (n + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t ch = 0; ch < nchunks; ch++) {
    const size_t i0 = ch * CH, i1 = (i0 + CH < n) ? i0 + CH : n;
    G(_jac) acc[CH];
    F(_t) pref[CH];
    for (size_t i = i0; i < i1; i++) {
      G(_jac)* a = &acc[i - i0];
      G(_set_inf)(a, P);
      for (int w = 0; w < NW; w++) {
        const int bit = w * WB;
        const size_t d = (size_t)((scalars[4 * i + (bit >> 6)] >> (bit & 63)) & (TW - 1));
        if (d) G(_madd)(a, a, &table[(size_t)w * TW + d], P);
      }
    }
    G(_normalize_chunk)(&out[i0], acc, i1 - i0, pref, P);
  }
  free(table);
}

#undef F
#undef G
#undef G2
#undef G2_
