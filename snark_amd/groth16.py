"""Host-side mirror of the reference's trait surface for the Groth16 path, over the C ABI.

Mirrors ``ark_snark::SNARK`` (/root/reference/snark/src/lib.rs:22-81) for one implementor,
``Groth16``:

* ``circuit_specific_setup``  (lib.rs:43-46, :87-92)  -> host computes the QAP scalars
  (u_i(tau), v_i(tau), w_i(tau), ...; SURVEY.md Appendix A "Setup"), the device performs the
  fixed-base multiplications (``ark355_fixed_base_mul``);
* ``prove``                   (lib.rs:50-54)           -> ``ark355_prove`` (the hot path);
* ``verify`` / ``process_vk`` (lib.rs:59-80)           -> ``ark355_verify_batch`` (random linear combination on
  the device MSM, Miller loops and the final exponentiation on host threads); proofs stay byte-compatible
  with the arkworks CPU verifier.

Inputs come from the unchanged ``ark-relations`` constraint system on the host:
``R1CS.from_rows`` takes exactly what ``ConstraintSystem::to_matrices()["R1CS"]`` returns
(/root/reference/relations/src/gr1cs/constraint_system.rs:768-774; ``Matrix<F> = Vec<Vec<(F, usize)>>``,
utils/matrix.rs:4) and ``z`` is ``instance_assignment || witness_assignment``
(constraint_system.rs:193-206).  Error behaviour follows ``SynthesisError`` (utils/error.rs:5-21).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lib as _lib
from ._binding import Ark355Error
from .params import Curve, CURVES


class SynthesisError(Exception):
    """Mirror of ark_relations::utils::error::SynthesisError (utils/error.rs:5-21)."""


@dataclass
class R1CS:
    """CSR image of the three R1CS matrices (column convention utils/variable.rs:105-113)."""
    curve: Curve
    n: int
    ell: int
    w: int
    row_ptr: Tuple[np.ndarray, np.ndarray, np.ndarray]
    col: Tuple[np.ndarray, np.ndarray, np.ndarray]
    coeff: Tuple[bytes, bytes, bytes]          # Montgomery Fr images, 32 B per non-zero
    coeff_int: Optional[Tuple[list, list, list]] = None   # canonical ints (setup needs them)

    @staticmethod
    def from_rows(curve: Curve, A, B, C, ell: int, w: int) -> "R1CS":
        """A, B, C: Vec<Vec<(F, usize)>> with F as canonical Python ints."""
        rps, cols, cfs, cis = [], [], [], []
        for M in (A, B, C):
            rp = np.zeros(len(M) + 1, dtype=np.uint64)
            cl, cf, ci = [], [], []
            k = 0
            for i, row in enumerate(M):
                for c, j in row:
                    cl.append(j)
                    ci.append(c % curve.r)
                    cf.append(curve.fr_mont(c))
                    k += 1
                rp[i + 1] = k
            rps.append(rp)
            cols.append(np.array(cl, dtype=np.uint32))
            cfs.append(b"".join(cf))
            cis.append(ci)
        return R1CS(curve, len(A), ell, w, tuple(rps), tuple(cols), tuple(cfs), tuple(cis))

    @property
    def m(self):
        return self.ell + self.w

    @property
    def domain_size(self):
        need = self.n + self.ell
        N = 1
        while N < need:
            N <<= 1
        return N


@dataclass
class VerifyingKey:
    alpha_g1: bytes
    beta_g2: bytes
    gamma_g2: bytes
    delta_g2: bytes
    gamma_abc_g1: bytes        # ell raw G1 points


@dataclass
class ProvingKey:
    vk: VerifyingKey
    beta_g1: bytes
    delta_g1: bytes
    a_query: bytes
    b_g1_query: bytes
    b_g2_query: bytes
    h_query: bytes
    l_query: bytes
    ell: int
    w: int
    N: int
    # retained only when setup is asked to keep the toxic waste (tests / benches: closed-form check)
    trapdoor: Optional[dict] = None

    def check_lengths(self, sizes):
        """The C ABI copies (ell+w), (N-1) and w points from bare pointers: every vector must really hold that many
        (a malformed serialized key must fail here, not read out of bounds)."""
        m, g1, g2 = self.ell + self.w, sizes["g1"], sizes["g2"]
        want = (("a_query", m * g1), ("b_g1_query", m * g1), ("b_g2_query", m * g2), ("h_query", (self.N - 1) * g1),
                ("l_query", self.w * g1), ("beta_g1", g1), ("delta_g1", g1))
        for name, n in want:
            if len(getattr(self, name)) != n:
                raise ValueError("proving key: %s holds %d bytes, expected %d" % (name, len(getattr(self, name)), n))
        for name, n in (("alpha_g1", g1), ("beta_g2", g2), ("delta_g2", g2)):
            if len(getattr(self.vk, name)) != n:
                raise ValueError("proving key: vk.%s holds %d bytes, expected %d" % (name, len(getattr(self.vk, name)), n))
        if self.ell < 1 or self.N < 1 or (self.N & (self.N - 1)):
            raise ValueError("proving key: bad dimensions")


@dataclass
class Proof:
    """Raw affine points (Montgomery images); `ark_groth16::Proof {a, b, c}`."""
    a: bytes
    b: bytes
    c: bytes


class Groth16:
    """`impl SNARK<Fr> for Groth16` over the MI355X backend."""

    def __init__(self, curve="bls12_381", device: int = 0, lib=None):
        self.curve: Curve = CURVES[curve] if not isinstance(curve, Curve) else curve
        # `lib` is injectable only so that tests can drive this host logic over the CPU emulator build;
        # the default (and only shipped) backend is the HIP library.
        self.lib = lib if lib is not None else _lib()
        self.ctx = self.lib.ctx_create(device)
        self.sizes = self.lib.sizes(self.curve.curve_id)
        self._finalizers = []

    # Device handles live ON the host object they image (attribute + weakref finalizer), never in a table keyed by
    # id(obj): CPython reuses ids after collection, and a dropped 2^20 key must give its ~15 GB of HBM back.
    def _track(self, obj, attr, value, free_fn, handle):
        import weakref
        setattr(obj, attr, value)
        fin = weakref.finalize(obj, free_fn, handle)
        fin.atexit = False
        self._finalizers.append(fin)
        self._finalizers = [f for f in self._finalizers if f.alive]

    def _cached(self, obj, attr):
        v = getattr(obj, attr, None)
        return v[1] if (v is not None and v[0] is self) else None

    def close(self):
        for fin in self._finalizers:
            fin()                                   # frees the handle once; a no-op if the object already died
        self._finalizers = []
        if self.ctx is not None:
            self.lib.ctx_destroy(self.ctx)
            self.ctx = None

    # ---- SNARK::circuit_specific_setup (snark/src/lib.rs:43-46) ---------------------------------------
    def circuit_specific_setup(self, r1cs: R1CS, rng, keep_trapdoor=False) -> Tuple[ProvingKey, VerifyingKey]:
        """`rng` yields field elements: callable rng() -> int (uniform mod r).  The generator's scalars (Lagrange
        coefficients at tau, u/v/w, l, gamma_abc, h) come from `ark355_setup_scalars` (host threads, the library's own
        field code), the query vectors from `ark355_fixed_base_mul` on the device."""
        cv, r = self.curve, self.curve.r
        tau, alpha, beta, gamma, delta = (rng() % r or 1 for _ in range(5))
        ell, m = r1cs.ell, r1cs.m
        N = r1cs.domain_size
        if N.bit_length() - 1 > cv.two_adicity:
            raise SynthesisError("PolynomialDegreeTooLarge")
        td = b"".join(cv.fr_canon(x) for x in (tau, alpha, beta, gamma, delta))
        try:
            sc = self.lib.setup_scalars(cv.curve_id, r1cs.n, ell, r1cs.w, list(zip(r1cs.row_ptr, r1cs.col, r1cs.coeff)), td)
        except Ark355Error as e:
            raise SynthesisError(str(e)) from e
        g1, g2 = cv.g1_gen_raw(), cv.g2_gen_raw()

        def mul(group, scalars, n=None):
            if not isinstance(scalars, np.ndarray):
                scalars = np.frombuffer(b"".join(cv.fr_canon(s) for s in scalars), dtype=np.uint8)
            n = len(scalars) // 32
            return self.lib.fixed_base_mul(self.ctx, cv.curve_id, group, g1 if group == 1 else g2, scalars, n,
                                           self.sizes["g1"] if group == 1 else self.sizes["g2"])

        singles1 = mul(1, [alpha, beta, delta])
        singles2 = mul(2, [beta, gamma, delta])
        s1, s2 = self.sizes["g1"], self.sizes["g2"]
        vk = VerifyingKey(alpha_g1=singles1[:s1], beta_g2=singles2[:s2], gamma_g2=singles2[s2:2 * s2],
                          delta_g2=singles2[2 * s2:], gamma_abc_g1=mul(1, sc["gamma_abc"]))
        pk = ProvingKey(vk=vk, beta_g1=singles1[s1:2 * s1], delta_g1=singles1[2 * s1:],
                        a_query=mul(1, sc["u"]), b_g1_query=mul(1, sc["v"]), b_g2_query=mul(2, sc["v"]),
                        h_query=mul(1, sc["h"]), l_query=mul(1, sc["l"]), ell=ell, w=r1cs.w, N=N)
        if keep_trapdoor:
            # u, v, w stay as canonical byte images; prove_closed_form turns them into integers on first use
            pk.trapdoor = dict(tau=tau, alpha=alpha, beta=beta, gamma=gamma, delta=delta,
                               u=sc["u"].tobytes(), v=sc["v"].tobytes(), w=sc["w"].tobytes())
        return pk, vk

    # ---- device residency ---------------------------------------------------------------------------------
    def load_pk(self, pk: ProvingKey):
        h = self._cached(pk, "_ark355_pk")
        if h is None:
            pk.check_lengths(self.sizes)
            h = self.lib.pk_load(self.ctx, self.curve.curve_id, pk.ell, pk.w, pk.N, pk.a_query, pk.b_g1_query,
                                 pk.b_g2_query, pk.h_query, pk.l_query, pk.vk.alpha_g1, pk.beta_g1, pk.delta_g1,
                                 pk.vk.beta_g2, pk.vk.delta_g2)
            self._track(pk, "_ark355_pk", (self, h), self.lib.dll.ark355_pk_free, h)
        return h

    def load_r1cs(self, r1cs: R1CS):
        h = self._cached(r1cs, "_ark355_r1cs")
        if h is None:
            try:
                h = self.lib.r1cs_load(self.ctx, self.curve.curve_id, r1cs.n, r1cs.ell, r1cs.w,
                                       list(zip(r1cs.row_ptr, r1cs.col, r1cs.coeff)))
            except Ark355Error as e:
                raise SynthesisError(str(e)) from e
            self._track(r1cs, "_ark355_r1cs", (self, h), self.lib.dll.ark355_r1cs_free, h)
        return h

    # ---- SNARK::prove (snark/src/lib.rs:50-54) ---------------------------------------------------------------
    def prove(self, pk: ProvingKey, r1cs: R1CS, z, rng=None, r: Optional[int] = None, s: Optional[int] = None,
              z_device_ptr: Optional[int] = None) -> Proof:
        """z: Montgomery bytes of the full assignment (or a list of ints).  r, s: the zero-knowledge
        randomisers -- drawn from `rng` in the order r then s (as upstream `create_random_proof`) unless given."""
        cv = self.curve
        if r is None:
            r = rng() % cv.r
        if s is None:
            s = rng() % cv.r
        pkh, rh = self.load_pk(pk), self.load_r1cs(r1cs)
        try:
            if z_device_ptr is not None:
                # The library reads z on its own streams: whatever produced it (e.g. torch's current stream) must have
                # finished.  With torch loaded this is one device synchronisation; other producers synchronise
                # themselves (include/ark355.h, "*_dev variants").
                import sys
                torch = sys.modules.get("torch")
                if torch is not None and torch.cuda.is_available():
                    torch.cuda.synchronize()
                a, b, c = self.lib.prove(self.ctx, pkh, rh, z_device_ptr, r1cs.m, cv.fr_canon(r), cv.fr_canon(s),
                                         self.sizes, z_is_device_ptr=True)
            else:
                if not isinstance(z, (bytes, bytearray, np.ndarray)):
                    z = b"".join(cv.fr_mont(v) for v in z)
                z_len = len(z) // 32
                a, b, c = self.lib.prove(self.ctx, pkh, rh, z, z_len, cv.fr_canon(r), cv.fr_canon(s), self.sizes)
        except Ark355Error as e:
            raise SynthesisError(str(e)) from e
        return Proof(a, b, c)

    def prove_batch(self, pk: ProvingKey, r1cs: R1CS, zs, rng=None, rs=None, inflight: int = 3):
        """Many proofs of one circuit (`ark355_prove_batch`): zs = one assignment per proof (Montgomery bytes or
        lists of ints); randomisers are drawn from `rng` in the order r_0, s_0, r_1, s_1, ... unless `rs` gives the
        (r, s) pairs.  Up to `inflight` proofs share the GPU at a time."""
        cv = self.curve
        if rs is None:
            rs = [(rng() % cv.r, rng() % cv.r) for _ in zs]
        zb = [z if isinstance(z, (bytes, bytearray, np.ndarray)) else b"".join(cv.fr_mont(v) for v in z) for z in zs]
        z_len = min((len(z) // 32 for z in zb), default=0)
        pkh, rh = self.load_pk(pk), self.load_r1cs(r1cs)
        try:
            out = self.lib.prove_batch(self.ctx, pkh, rh, zb, z_len, [cv.fr_canon(r) for r, _ in rs],
                                       [cv.fr_canon(s) for _, s in rs], self.sizes, inflight=inflight)
        except Ark355Error as e:
            raise SynthesisError(str(e)) from e
        return [Proof(a, b, c) for a, b, c in out]

    def is_satisfied(self, r1cs: R1CS, z) -> Optional[int]:
        """None if satisfied, else the first failing constraint index (constraint_system.rs:661-687)."""
        cv = self.curve
        if not isinstance(z, (bytes, bytearray, np.ndarray)):
            z = b"".join(cv.fr_mont(v) for v in z)
        fb = self.lib.is_satisfied(self.ctx, self.load_r1cs(r1cs), z, len(z) // 32)
        return None if fb < 0 else fb

    # ---- closed form from a retained trapdoor (bench / tests: size-independent proof check) -----------------------
    def prove_closed_form(self, pk: ProvingKey, z_ints: Sequence[int], r: int, s: int) -> Proof:
        td = pk.trapdoor
        if td is None:
            raise ValueError("setup was not asked to keep the trapdoor")
        cv, R = self.curve, self.curve.r
        m, ell = len(z_ints), pk.ell
        for k in "uvw":                      # canonical byte images -> integers, once
            if isinstance(td[k], (bytes, bytearray)):
                b = td[k]
                td[k] = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(len(b) // 32)]
        # the four sums over z do not depend on (r, s): kept for the assignment they were computed for (a bench checks
        # dozens of proofs of ONE assignment; 0.6 s each at n = 2^20 otherwise)
        cache = td.get("_sums")
        if cache is None or cache[0] is not z_ints:
            az = sum(z_ints[i] * td["u"][i] for i in range(m)) % R
            bz = sum(z_ints[i] * td["v"][i] for i in range(m)) % R
            cz = sum(z_ints[i] * td["w"][i] for i in range(m)) % R
            l_part = sum(z_ints[i] * (td["beta"] * td["u"][i] + td["alpha"] * td["v"][i] + td["w"][i])
                         for i in range(ell, m)) % R
            td["_sums"] = cache = (z_ints, az, bz, cz, l_part)
        _, az, bz, cz, l_part = cache
        a_exp = (td["alpha"] + az + r * td["delta"]) % R
        b_exp = (td["beta"] + bz + s * td["delta"]) % R
        di = pow(td["delta"], -1, R)
        c_exp = ((l_part + az * bz - cz) * di + s * a_exp + r * b_exp - r * s % R * td["delta"]) % R
        g1 = self.lib.fixed_base_mul(self.ctx, cv.curve_id, 1, cv.g1_gen_raw(),
                                     cv.fr_canon(a_exp) + cv.fr_canon(c_exp), 2, self.sizes["g1"])
        g2 = self.lib.fixed_base_mul(self.ctx, cv.curve_id, 2, cv.g2_gen_raw(), cv.fr_canon(b_exp), 1,
                                     self.sizes["g2"])
        s1 = self.sizes["g1"]
        return Proof(g1[:s1], g2, g1[s1:])

    # ---- SNARK::verify (snark/src/lib.rs:59-80) -----------------------------------------------------------------------
    def verify(self, vk: VerifyingKey, public_inputs: Sequence[int], proof: Proof) -> bool:
        """`public_inputs` excludes the leading One, as in `SNARK::verify`.  One proof = a batch of one."""
        return self.verify_batch(vk, [public_inputs], [proof])

    def process_vk(self, vk: VerifyingKey) -> VerifyingKey:
        """Nothing is precomputed on the host side of this backend: the processed key is the key (lib.rs:70-73)."""
        return vk

    verify_with_processed_vk = verify

    def verify_batch(self, vk: VerifyingKey, public_inputs, proofs, rng=None) -> bool:
        """`ark355_verify_batch`: all proofs of ONE verifying key checked with a random linear combination (count + 3
        Miller loops, one final exponentiation).  `rng` yields the 128-bit coefficients; required for more than one proof."""
        cv = self.curve
        count = len(proofs)
        if count == 0:
            return True
        ell = len(vk.gamma_abc_g1) // self.sizes["g1"]
        if any(len(x) + 1 != ell for x in public_inputs) or len(public_inputs) != count:
            return False                          # upstream: MalformedVerifyingKey / wrong input length -> not accepted
        rho = None
        if count > 1:
            if rng is None:
                raise ValueError("batch verification needs an rng for the random coefficients")
            rho = [cv.fr_canon((rng() % ((1 << 128) - 1)) + 1) for _ in range(count)]
        xs = b"".join(cv.fr_mont(v) for row in public_inputs for v in row)
        try:
            return self.lib.verify_batch(self.ctx, cv.curve_id,
                                         (vk.alpha_g1, vk.beta_g2, vk.gamma_g2, vk.delta_g2, vk.gamma_abc_g1),
                                         [(p.a, p.b, p.c) for p in proofs], xs, rho)
        except Ark355Error as e:
            raise SynthesisError(str(e)) from e
