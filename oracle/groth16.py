"""Groth16 over an R1CS instance -- step-by-step oracle (O1) and trapdoor closed form (O2).

Restates the published algorithm of the un-vendored ``ark-groth16`` crate
(``src/{generator,r1cs_to_qap,prover,verifier}.rs``; SURVEY.md Appendix A is the normative
spec) behind the trait ``ark_snark::SNARK`` (/root/reference/snark/src/lib.rs:22-81):
``circuit_specific_setup`` (:43-46), ``prove`` (:50-54), ``verify`` (:59-80).

Test infrastructure only.  PARITY UNPINNED by the reference (it holds no proof vectors);
pinned here by O1 == O2 byte equality + the pairing equation.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

from .fields import CurveParams
from .curves import g1 as G1of, g2 as G2of
from .ntt import Domain
from .pairing import Pairing


@dataclass
class Trapdoor:
    tau: int
    alpha: int
    beta: int
    gamma: int
    delta: int
    g1_k: int = 1      # G1 generator used = g1_k * standard generator
    g2_k: int = 1


@dataclass
class VerifyingKey:
    alpha_g1: tuple
    beta_g2: tuple
    gamma_g2: tuple
    delta_g2: tuple
    gamma_abc_g1: list


@dataclass
class ProvingKey:
    vk: VerifyingKey
    beta_g1: tuple
    delta_g1: tuple
    a_query: list
    b_g1_query: list
    b_g2_query: list
    h_query: list
    l_query: list
    # oracle-only extras (retained trapdoor data for the closed form)
    trapdoor: Optional[Trapdoor] = None
    u: list = field(default_factory=list)
    v: list = field(default_factory=list)
    w: list = field(default_factory=list)
    domain_log: int = 0


@dataclass
class Proof:
    a: tuple
    b: tuple
    c: tuple


def qap_scalars(curve: CurveParams, A, B, C, n_constraints, ell, m, tau):
    """u_i(tau), v_i(tau), w_i(tau), Z(tau), domain (R1CSToQAP::instance_map_with_evaluation)."""
    r = curve.r
    dom = Domain.for_size(curve, n_constraints + ell)
    L = dom.lagrange_at(tau)
    u = [0] * m
    v = [0] * m
    w = [0] * m
    for i in range(ell):
        u[i] = L[n_constraints + i]
    for k in range(n_constraints):
        lk = L[k]
        for c, j in A[k]:
            u[j] = (u[j] + lk * c) % r
        for c, j in B[k]:
            v[j] = (v[j] + lk * c) % r
        for c, j in C[k]:
            w[j] = (w[j] + lk * c) % r
    zt = (pow(tau, dom.n, r) - 1) % r
    return u, v, w, zt, dom


def setup(curve: CurveParams, A, B, C, ell, m, td: Trapdoor, fast=True) -> ProvingKey:
    """Groth16 generator with an explicit (retained) trapdoor."""
    r = curve.r
    n = len(A)
    G1, G2 = G1of(curve), G2of(curve)
    g1 = G1.mul(curve.g1_gen, td.g1_k)
    g2 = G2.mul(curve.g2_gen, td.g2_k)
    u, v, w, zt, dom = qap_scalars(curve, A, B, C, n, ell, m, td.tau)
    N = dom.n
    gi = pow(td.gamma, -1, r)
    di = pow(td.delta, -1, r)
    abc = [(td.beta * u[i] + td.alpha * v[i] + w[i]) % r for i in range(m)]
    gamma_abc_s = [abc[i] * gi % r for i in range(ell)]
    l_s = [abc[i] * di % r for i in range(ell, m)]
    h_s = []
    t = zt * di % r
    for _ in range(N - 1):
        h_s.append(t)
        t = t * td.tau % r
    mul1 = (lambda ks: G1.fixed_base_muls(g1, ks)) if fast else (lambda ks: [G1.mul(g1, k) for k in ks])
    mul2 = (lambda ks: G2.fixed_base_muls(g2, ks)) if fast else (lambda ks: [G2.mul(g2, k) for k in ks])
    vk = VerifyingKey(
        alpha_g1=G1.mul(g1, td.alpha), beta_g2=G2.mul(g2, td.beta),
        gamma_g2=G2.mul(g2, td.gamma), delta_g2=G2.mul(g2, td.delta),
        gamma_abc_g1=mul1(gamma_abc_s))
    return ProvingKey(
        vk=vk, beta_g1=G1.mul(g1, td.beta), delta_g1=G1.mul(g1, td.delta),
        a_query=mul1(u), b_g1_query=mul1(v), b_g2_query=mul2(v),
        h_query=mul1(h_s), l_query=mul1(l_s),
        trapdoor=td, u=u, v=v, w=w, domain_log=dom.log_n)


def witness_map(curve: CurveParams, A, B, C, z, ell):
    """h[0..N) (LibsnarkReduction::witness_map_from_matrices)."""
    r = curve.r
    n = len(A)
    dom = Domain.for_size(curve, n + ell)
    N = dom.n
    a = [0] * N
    b = [0] * N
    c = [0] * N
    for i in range(n):
        a[i] = sum(co * z[j] for co, j in A[i]) % r
        b[i] = sum(co * z[j] for co, j in B[i]) % r
        c[i] = sum(co * z[j] for co, j in C[i]) % r
    for j in range(ell):
        a[n + j] = z[j] % r
    a = dom.coset_fft(dom.ifft(a))
    b = dom.coset_fft(dom.ifft(b))
    c = dom.coset_fft(dom.ifft(c))
    zi = dom.vanishing_on_coset_inv()
    t = [((a[i] * b[i] - c[i]) * zi) % r for i in range(N)]
    return dom.coset_ifft(t)


def prove(curve: CurveParams, pk: ProvingKey, A, B, C, z, ell, r_, s_, h=None) -> Proof:
    """create_proof_with_reduction_and_matrices (O1: step by step, naive MSMs)."""
    G1, G2 = G1of(curve), G2of(curve)
    R = curve.r
    m = len(z)
    if h is None:
        h = witness_map(curve, A, B, C, z, ell)
    h_acc = G1.msm(pk.h_query, h[:len(pk.h_query)])
    l_acc = G1.msm(pk.l_query, z[ell:])
    assert len(pk.a_query) == m and len(pk.l_query) == m - ell

    def coeff(G, initial, query, vk_param):
        acc = G.msm(query[1:], z[1:])
        return G.sum([initial, query[0], acc, vk_param])

    g_a = coeff(G1, G1.mul(pk.delta_g1, r_), pk.a_query, pk.vk.alpha_g1)
    g1_b = coeff(G1, G1.mul(pk.delta_g1, s_), pk.b_g1_query, pk.beta_g1) if r_ % R else None
    g2_b = coeff(G2, G2.mul(pk.vk.delta_g2, s_), pk.b_g2_query, pk.vk.beta_g2)
    g_c = G1.sum([G1.mul(g_a, s_), G1.mul(g1_b, r_) if g1_b else None,
                  G1.neg(G1.mul(pk.delta_g1, r_ * s_ % R)), l_acc, h_acc])
    return Proof(a=g_a, b=g2_b, c=g_c)


def prove_closed_form(curve: CurveParams, pk: ProvingKey, z, ell, r_, s_) -> Proof:
    """O2: exponents from the retained trapdoor; no MSM, no FFT.  Valid for satisfying z."""
    td = pk.trapdoor
    assert td is not None
    R = curve.r
    G1, G2 = G1of(curve), G2of(curve)
    g1 = G1.mul(curve.g1_gen, td.g1_k)
    g2 = G2.mul(curve.g2_gen, td.g2_k)
    m = len(z)
    az = sum(z[i] * pk.u[i] for i in range(m)) % R
    bz = sum(z[i] * pk.v[i] for i in range(m)) % R
    cz = sum(z[i] * pk.w[i] for i in range(m)) % R
    a_exp = (td.alpha + az + r_ * td.delta) % R
    b_exp = (td.beta + bz + s_ * td.delta) % R
    di = pow(td.delta, -1, R)
    l_part = sum(z[i] * (td.beta * pk.u[i] + td.alpha * pk.v[i] + pk.w[i]) for i in range(ell, m)) % R
    hz = (az * bz - cz) % R                       # = h(tau) * Z(tau)
    c_exp = ((l_part + hz) * di + s_ * a_exp + r_ * b_exp - r_ * s_ % R * td.delta) % R
    return Proof(a=G1.mul(g1, a_exp), b=G2.mul(g2, b_exp), c=G1.mul(g1, c_exp))


_pairings = {}


def verify(curve: CurveParams, vk: VerifyingKey, public_inputs: List[int], proof: Proof) -> bool:
    """e(A,B) == e(alpha,beta) e(sum x_i gamma_abc_i, gamma) e(C,delta); public_inputs excludes the 1."""
    G1 = G1of(curve)
    if len(public_inputs) + 1 != len(vk.gamma_abc_g1):
        return False
    if curve.name not in _pairings:
        _pairings[curve.name] = Pairing(curve)
    P = _pairings[curve.name]
    acc = G1.msm(vk.gamma_abc_g1, [1] + list(public_inputs))
    return P.pairing_product_is_one([
        (proof.a, proof.b),
        (G1.neg(vk.alpha_g1), vk.beta_g2),
        (G1.neg(acc), vk.gamma_g2),
        (G1.neg(proof.c), vk.delta_g2),
    ])
