//! `ark-mi355x`: the arkworks `SNARK` trait surface (snark/src/lib.rs:22-93) over libark355.so, the MI355X-native
//! Groth16 prover.  Key, proof and verifying-key types are `ark_groth16`'s own, so bytes are interchangeable with the CPU
//! prover; setup and verification stay on the CPU implementation, `prove` runs on the GPU.
//!
//! UNCOMPILED in the repository's own build environment (it has no Rust toolchain); the C ABI underneath is exercised
//! there through the C++ and Python mirrors.  `tests/parity.rs` is the test to run on a machine with both toolchains.
mod cache;
mod error;
pub mod ffi;
mod marshal;
mod pool;

use core::marker::PhantomData;

use ark_ec::short_weierstrass::{Affine, SWCurveConfig};
use ark_ff::{Field, PrimeField, UniformRand};
use ark_groth16::{Groth16, PreparedVerifyingKey, Proof, ProvingKey, VerifyingKey};
use ark_relations::gr1cs::{
    ConstraintSynthesizer, ConstraintSystem, ConstraintSystemRef, OptimizationGoal, SynthesisError, SynthesisMode,
    R1CS_PREDICATE_LABEL,
};
use ark_snark::{CircuitSpecificSetupSNARK, SNARK};
use ark_std::rand::{CryptoRng, RngCore};

pub use cache::{evict, lookup, matrices_hash, set_device, trust_cache, Resident};
pub use error::Mi355xError;
pub use marshal::{layout_self_test, Mi355xCurve};
pub use pool::{PinnedAssignment, ProverPool, Ticket};

/// Drop-in sibling of `ark_groth16::Groth16<E>`.
pub struct Mi355xGroth16<E>(PhantomData<E>);

/// Everything `prove` needs from one synthesis run.
struct Synthesized<F: PrimeField> {
    cs: ConstraintSystemRef<F>,
    z: Vec<F>,
}

/// What `prepare` hands to the proving call: the resident handles, the full assignment, whether the witness-only fast path
/// produced it, and the circuit type (for `refused`).
struct Prepared<F: PrimeField> {
    res: std::sync::Arc<cache::Resident>,
    z: Vec<F>,
    fast_path: bool,
    tname: &'static str,
}

/// Run the circuit.  With the matrices already resident the constraint system is put in witness-only mode
/// (`construct_matrices: false`: `enforce_r1cs_constraint` returns without recording anything,
/// relations/src/gr1cs/constraint_system_ref.rs:241-243), which is what every proof after the first pays.
fn synthesize<F: PrimeField, C: ConstraintSynthesizer<F>>(circuit: C, witness_only: bool) -> Result<Synthesized<F>, SynthesisError> {
    let cs = ConstraintSystem::<F>::new_ref(); // constraint_system.rs:142
    cs.set_optimization_goal(OptimizationGoal::Constraints); // :563
    if witness_only {
        cs.set_mode(SynthesisMode::Prove { construct_matrices: false, generate_lc_assignments: false });
    }
    circuit.generate_constraints(cs.clone())?; // gr1cs/mod.rs:60
    cs.finalize(); // constraint_system_ref.rs:435
    let mut z = cs.instance_assignment()?; // instance || witness, constraint_system.rs:193-206
    z.extend(cs.witness_assignment()?);
    Ok(Synthesized { cs, z })
}

impl<E, P1, P2> Mi355xGroth16<E>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    /// Synthesis + residency for one proof: the handles, the full assignment, and whether the witness-only fast path
    /// produced it (`Prepared`).
    ///
    /// What a caller observes does not depend on the state of the cache.  Every proof runs with the library's
    /// policy `CHECK_SATISFIED` (set when the thread's context is created, `cache::with_ctx`): the rows `a_i b_i = c_i` are
    /// compared on the device as part of the witness map, at no extra copy of the assignment, and an assignment the matrices
    /// do not accept comes back as `SynthesisError::Unsatisfiable` -- on the first call of a key (full path: the matrices are
    /// the caller's own) exactly as on every later one (fast path: the matrices are the cached entry's).  That is the check
    /// `ark-groth16` makes under `debug_assert!(cs.is_satisfied())`; its release build returns a proof that cannot verify
    /// instead, and `ARK_MI355X_TRUST_CACHE=1` restores that behaviour (no check at all) for hosts that prove one circuit
    /// per key.
    fn prepare<C: ConstraintSynthesizer<E::ScalarField>>(pk: &ProvingKey<E>, circuit: C) -> Result<Prepared<E::ScalarField>, Mi355xError> {
        // Entries are found by the key's SAMPLED fingerprint, which two circuits can share (cache::fingerprint): an entry is
        // used for the witness-only fast path only by circuit types whose matrices were compared with it once and that have
        // never been caught building a different system afterwards (`cache::is_unstable`).
        let tname = core::any::type_name::<C>();
        let cands = cache::candidates(pk);
        if !cache::is_unstable(pk, tname) {
            if let Some(res) = cands.iter().find(|r| r.is_confirmed(tname)) {
                let syn = synthesize(circuit, true)?;
                if syn.z.len() == res.num_instance + res.num_witness {
                    cache::touch(pk, res);
                    return Ok(Prepared { res: res.clone(), z: syn.z, fast_path: true, tname });
                }
                // A different number of variables: this type builds more than one system.  The circuit is consumed, so this
                // proof cannot be redone here; from now on the type takes the full path.
                cache::mark_unstable(pk, tname);
                res.unconfirm(tname);
                return Err(Mi355xError::Synthesis(SynthesisError::Unsatisfiable));
            }
        }
        let syn = synthesize(circuit, false)?;
        let mats = syn.cs.to_matrices()?; // constraint_system.rs:768
        let r1cs = mats.get(R1CS_PREDICATE_LABEL).ok_or(Mi355xError::Synthesis(SynthesisError::PredicateNotFound))?;
        let mh = cache::matrices_hash(r1cs, syn.cs.num_constraints());
        if let Some(res) = cands.iter().find(|r| r.matrices_hash == mh) {
            res.confirm(tname);
            cache::touch(pk, res);
            return Ok(Prepared { res: res.clone(), z: syn.z, fast_path: false, tname });
        }
        let res = cache::load::<E, P1, P2>(
            pk,
            r1cs,
            syn.cs.num_constraints(),
            syn.cs.num_instance_variables(),
            syn.cs.num_witness_variables(),
        )?;
        res.confirm(tname);
        Ok(Prepared { res, z: syn.z, fast_path: false, tname })
    }

    /// The library refused an assignment (`CHECK_SATISFIED`).  On the fast path the shim cannot tell an unsatisfied circuit
    /// from a circuit TYPE that builds different systems from call to call (the circuit has been consumed in witness-only
    /// mode): the type loses the fast path for this key for good -- its later proofs rebuild and compare the matrices -- so a
    /// type that alternates between two systems fails at most once, and a genuinely unsatisfied circuit fails every time,
    /// cached or not.
    fn refused(pk: &ProvingKey<E>, prep: &Prepared<E::ScalarField>, e: Mi355xError) -> Mi355xError {
        if prep.fast_path && matches!(e, Mi355xError::Synthesis(SynthesisError::Unsatisfiable)) {
            cache::mark_unstable(pk, prep.tname);
            prep.res.unconfirm(prep.tname);
        }
        e
    }

    /// `create_proof_with_reduction`'s counterpart: explicit zero-knowledge randomisers (parity tests).
    pub fn prove_with_rs<C: ConstraintSynthesizer<E::ScalarField>>(
        pk: &ProvingKey<E>,
        circuit: C,
        r: E::ScalarField,
        s: E::ScalarField,
    ) -> Result<Proof<E>, Mi355xError> {
        let prep = Self::prepare(pk, circuit)?;
        let (rb, sb) = (marshal::canonical_32(&r), marshal::canonical_32(&s));
        let mut raw = ffi::ark355_proof_raw { a: [0; 96], b: [0; 192], c: [0; 96] };
        cache::with_ctx(|ctx| {
            let zi = marshal::scalars_image(&prep.z);
            cache::check(ctx, unsafe {
                ffi::ark355_prove(ctx, prep.res.pk, prep.res.r1cs, zi.as_ptr(), prep.z.len() as u64, rb.as_ptr(), sb.as_ptr(), &mut raw)
            })
        })
        .map_err(|e| Self::refused(pk, &prep, e))?;
        Ok(marshal::proof_from_raw::<E, P1, P2>(&raw))
    }

    /// Many proofs of ONE circuit (`ark355_prove_batch`; BASELINE configs[4]).  Synthesis stays one circuit at a time
    /// on the calling thread (`ConstraintSystemRef` is `!Send`); callers that want it overlapped run this from several
    /// threads or synthesise on a pool and call `prove_assignments`.
    pub fn prove_batch<C: ConstraintSynthesizer<E::ScalarField>, R: RngCore + CryptoRng>(
        pk: &ProvingKey<E>,
        circuits: Vec<C>,
        rng: &mut R,
        inflight: u32,
    ) -> Result<Vec<Proof<E>>, Mi355xError> {
        let mut zs = Vec::with_capacity(circuits.len());
        let mut last: Option<Prepared<E::ScalarField>> = None;
        let mut any_fast = false;
        for c in circuits {
            let mut prep = Self::prepare(pk, c)?;
            if let Some(prev) = &last {
                if !std::sync::Arc::ptr_eq(&prev.res, &prep.res) {
                    return Err(Mi355xError::InvalidArgument("prove_batch: the circuits do not share one constraint system".into()));
                }
            }
            any_fast |= prep.fast_path;
            zs.push(core::mem::take(&mut prep.z));
            last = Some(prep);
        }
        match last {
            None => Ok(Vec::new()),
            Some(mut prep) => {
                prep.fast_path = any_fast;
                Self::prove_assignments(&prep.res, &zs, rng, inflight).map_err(|e| Self::refused(pk, &prep, e))
            }
        }
    }

    /// Batch verification (`ark355_verify_batch`): every proof of ONE verifying key checked with a random linear
    /// combination -- `proofs.len() + 3` Miller loops and one final exponentiation.  `public_inputs[j]` excludes the
    /// leading One, as in `SNARK::verify`.
    pub fn verify_batch<R: RngCore + CryptoRng>(
        vk: &VerifyingKey<E>,
        public_inputs: &[Vec<E::ScalarField>],
        proofs: &[Proof<E>],
        rng: &mut R,
    ) -> Result<bool, Mi355xError> {
        let ell = vk.gamma_abc_g1.len();
        if proofs.is_empty() {
            return Ok(true);
        }
        if public_inputs.len() != proofs.len() || public_inputs.iter().any(|x| x.len() + 1 != ell) {
            return Ok(false);
        }
        let one = |p: &Affine<P1>| marshal::flatten_points(core::slice::from_ref(p));
        let one2 = |p: &Affine<P2>| marshal::flatten_points(core::slice::from_ref(p));
        let (alpha, beta, gamma, delta) = (one(&vk.alpha_g1), one2(&vk.beta_g2), one2(&vk.gamma_g2), one2(&vk.delta_g2));
        let gabc = marshal::flatten_points(&vk.gamma_abc_g1);
        let desc = ffi::ark355_vk_desc {
            num_instance: ell as u64,
            alpha_g1: alpha.as_ptr(),
            beta_g2: beta.as_ptr(),
            gamma_g2: gamma.as_ptr(),
            delta_g2: delta.as_ptr(),
            gamma_abc_g1: gabc.as_ptr(),
        };
        let mut raw = Vec::with_capacity(proofs.len());
        for p in proofs {
            let mut r = ffi::ark355_proof_raw { a: [0; 96], b: [0; 192], c: [0; 96] };
            let (a, b, c) = (one(&p.a), one2(&p.b), one(&p.c));
            r.a[..a.len()].copy_from_slice(&a);
            r.b[..b.len()].copy_from_slice(&b);
            r.c[..c.len()].copy_from_slice(&c);
            raw.push(r);
        }
        let xs: Vec<E::ScalarField> = public_inputs.iter().flatten().copied().collect();
        let mut rho = Vec::with_capacity(32 * proofs.len());
        for _ in proofs {
            let mut k = [0u8; 32];
            rng.fill_bytes(&mut k[..16]); // 128-bit coefficients
            k[0] |= 1; // non-zero
            rho.extend_from_slice(&k);
        }
        let mut ok = 0i32;
        cache::with_ctx(|ctx| {
            cache::check(ctx, unsafe {
                ffi::ark355_verify_batch(
                    ctx,
                    E::CURVE_ID,
                    &desc,
                    raw.as_ptr(),
                    marshal::scalars_image(&xs).as_ptr(),
                    if proofs.len() > 1 { rho.as_ptr() } else { core::ptr::null() },
                    proofs.len() as u64,
                    &mut ok,
                )
            })
        })?;
        Ok(ok == 1)
    }

    /// Assignments already synthesised (each `instance || witness`): up to `inflight` proofs share the GPU.
    pub fn prove_assignments<R: RngCore + CryptoRng>(
        res: &cache::Resident,
        zs: &[Vec<E::ScalarField>],
        rng: &mut R,
        inflight: u32,
    ) -> Result<Vec<Proof<E>>, Mi355xError> {
        let m = res.num_instance + res.num_witness;
        if zs.iter().any(|z| z.len() != m) {
            return Err(Mi355xError::Synthesis(SynthesisError::AssignmentMissing));
        }
        let (mut rs, mut ss) = (Vec::with_capacity(32 * zs.len()), Vec::with_capacity(32 * zs.len()));
        for _ in zs {
            // order: r then s, as upstream create_random_proof
            rs.extend_from_slice(&marshal::canonical_32(&E::ScalarField::rand(rng)));
            ss.extend_from_slice(&marshal::canonical_32(&E::ScalarField::rand(rng)));
        }
        let ptrs: Vec<*const u8> = zs.iter().map(|z| marshal::scalars_image(z).as_ptr()).collect();
        let mut raw = vec![ffi::ark355_proof_raw { a: [0; 96], b: [0; 192], c: [0; 96] }; zs.len()];
        cache::with_ctx(|ctx| {
            cache::check(ctx, unsafe {
                ffi::ark355_prove_batch(
                    ctx,
                    res.pk,
                    res.r1cs,
                    ptrs.as_ptr(),
                    m as u64,
                    rs.as_ptr(),
                    ss.as_ptr(),
                    zs.len() as u64,
                    inflight.clamp(1, 16),
                    raw.as_mut_ptr(),
                )
            })
        })?;
        Ok(raw.iter().map(marshal::proof_from_raw::<E, P1, P2>).collect())
    }
}

/// One proof with its MSM term ranges sharded over the GPUs of a node (BASELINE configs[2]): one process (or thread
/// with its own device) per rank.  `id` comes from `comm_unique_id()` on rank 0 and travels over the host's own IPC.
#[cfg(feature = "multi-gpu")]
pub mod sharded {
    use super::*;

    pub use crate::ffi::{ARK355_SHARD_BUCKET_RING, ARK355_SHARD_WINDOW};

    pub fn comm_unique_id() -> Result<[u8; ffi::ARK355_COMM_ID_BYTES], Mi355xError> {
        let mut id = [0u8; ffi::ARK355_COMM_ID_BYTES];
        let rc = unsafe { ffi::ark355_comm_unique_id(id.as_mut_ptr()) };
        if rc != ffi::ARK355_OK {
            return Err(Mi355xError::from_code(rc, "ark355_comm_unique_id".into()));
        }
        Ok(id)
    }

    /// This rank's communicator, key shard and matrices.
    pub struct ShardedProver<E> {
        comm: *mut ffi::ark355_comm,
        pk: *mut ffi::ark355_pk,
        r1cs: *mut ffi::ark355_r1cs,
        m: usize,
        _e: PhantomData<E>,
    }

    impl<E> Drop for ShardedProver<E> {
        fn drop(&mut self) {
            unsafe {
                ffi::ark355_comm_destroy(self.comm);
                ffi::ark355_pk_free(self.pk);
                ffi::ark355_r1cs_free(self.r1cs);
            }
        }
    }

    impl<E, P1, P2> ShardedProver<E>
    where
        E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
        P1: SWCurveConfig,
        P2: SWCurveConfig,
        <P1::BaseField as Field>::BasePrimeField: PrimeField,
        <P2::BaseField as Field>::BasePrimeField: PrimeField,
    {
        /// Collective over all ranks (like `ncclCommInitRank`).  `circuit` is synthesised once to obtain the matrices.
        pub fn new<C: ConstraintSynthesizer<E::ScalarField>>(
            id: &[u8; ffi::ARK355_COMM_ID_BYTES],
            rank: i32,
            world: i32,
            pk: &ProvingKey<E>,
            circuit: C,
        ) -> Result<Self, Mi355xError> {
            let syn = synthesize(circuit, false)?;
            let mats = syn.cs.to_matrices()?;
            let r1cs = mats.get(R1CS_PREDICATE_LABEL).ok_or(Mi355xError::Synthesis(SynthesisError::PredicateNotFound))?;
            let flat = marshal::flatten_key::<E, P1, P2>(pk)?;
            let csr: Vec<_> = r1cs.iter().map(marshal::csr_from_matrix).collect();
            cache::with_ctx(|ctx| {
                let mut comm = core::ptr::null_mut();
                cache::check(ctx, unsafe { ffi::ark355_comm_init(ctx, id.as_ptr(), rank, world, &mut comm) })?;
                let desc = flat.desc();
                let mut pk_h = core::ptr::null_mut();
                let rc = unsafe { ffi::ark355_pk_load_shard(ctx, E::CURVE_ID, &desc, rank as u32, world as u32, &mut pk_h) };
                if rc != ffi::ARK355_OK {
                    let e = Mi355xError::from_code(rc, cache::last_error(ctx));
                    unsafe { ffi::ark355_comm_destroy(comm) };
                    return Err(e);
                }
                let row_ptr = [csr[0].row_ptr.as_ptr(), csr[1].row_ptr.as_ptr(), csr[2].row_ptr.as_ptr()];
                let col = [csr[0].col.as_ptr(), csr[1].col.as_ptr(), csr[2].col.as_ptr()];
                let coeff = [
                    marshal::scalars_image(&csr[0].coeff).as_ptr(),
                    marshal::scalars_image(&csr[1].coeff).as_ptr(),
                    marshal::scalars_image(&csr[2].coeff).as_ptr(),
                ];
                let mut r1_h = core::ptr::null_mut();
                let rc = unsafe {
                    ffi::ark355_r1cs_load(
                        ctx,
                        E::CURVE_ID,
                        syn.cs.num_constraints() as u64,
                        syn.cs.num_instance_variables() as u64,
                        syn.cs.num_witness_variables() as u64,
                        row_ptr.as_ptr(),
                        col.as_ptr(),
                        coeff.as_ptr(),
                        &mut r1_h,
                    )
                };
                if rc != ffi::ARK355_OK {
                    let e = Mi355xError::from_code(rc, cache::last_error(ctx));
                    unsafe {
                        ffi::ark355_pk_free(pk_h);
                        ffi::ark355_comm_destroy(comm);
                    }
                    return Err(e);
                }
                Ok(Self { comm, pk: pk_h, r1cs: r1_h, m: syn.z.len(), _e: PhantomData })
            })
        }

        /// Collective: every rank passes the same circuit (same assignment) and the same `r`, `s`; every rank gets the
        /// same proof.  `mode`: `ARK355_SHARD_WINDOW` (all-gather of partial sums) or `ARK355_SHARD_BUCKET_RING`.
        pub fn prove<C: ConstraintSynthesizer<E::ScalarField>>(
            &self,
            circuit: C,
            r: E::ScalarField,
            s: E::ScalarField,
            mode: i32,
        ) -> Result<Proof<E>, Mi355xError> {
            let syn = synthesize(circuit, true)?;
            if syn.z.len() != self.m {
                return Err(Mi355xError::Synthesis(SynthesisError::AssignmentMissing));
            }
            let (rb, sb) = (marshal::canonical_32(&r), marshal::canonical_32(&s));
            let mut raw = ffi::ark355_proof_raw { a: [0; 96], b: [0; 192], c: [0; 96] };
            cache::with_ctx(|ctx| {
                let zi = marshal::scalars_image(&syn.z);
                cache::check(ctx, unsafe {
                    ffi::ark355_prove_sharded(
                        ctx,
                        self.comm,
                        self.pk,
                        self.r1cs,
                        zi.as_ptr(),
                        syn.z.len() as u64,
                        rb.as_ptr(),
                        sb.as_ptr(),
                        mode,
                        &mut raw,
                    )
                })
            })?;
            Ok(marshal::proof_from_raw::<E, P1, P2>(&raw))
        }
    }
}

impl<E, P1, P2> SNARK<E::ScalarField> for Mi355xGroth16<E>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    type ProvingKey = ProvingKey<E>; // snark/src/lib.rs:25
    type VerifyingKey = VerifyingKey<E>; // :29
    type Proof = Proof<E>; // :32
    type ProcessedVerifyingKey = PreparedVerifyingKey<E>; // :36
    type Error = Mi355xError; // :39 -- SynthesisError plus distinct device-side variants

    /// snark/src/lib.rs:43-46.  The generator is the CPU implementation (its five fixed-base loops can be handed to
    /// `ark355_fixed_base_mul`, as the Python / C++ mirrors of this repository do).
    fn circuit_specific_setup<C: ConstraintSynthesizer<E::ScalarField>, R: RngCore + CryptoRng>(
        circuit: C,
        rng: &mut R,
    ) -> Result<(Self::ProvingKey, Self::VerifyingKey), Self::Error> {
        Groth16::<E>::circuit_specific_setup(circuit, rng).map_err(Mi355xError::from)
    }

    /// snark/src/lib.rs:50-54 -- the accelerated path.
    fn prove<C: ConstraintSynthesizer<E::ScalarField>, R: RngCore + CryptoRng>(
        pk: &Self::ProvingKey,
        circuit: C,
        rng: &mut R,
    ) -> Result<Self::Proof, Self::Error> {
        // order: r then s, as upstream `create_random_proof`
        let r = E::ScalarField::rand(rng);
        let s = E::ScalarField::rand(rng);
        Self::prove_with_rs(pk, circuit, r, s)
    }

    /// snark/src/lib.rs:70-73
    fn process_vk(vk: &Self::VerifyingKey) -> Result<Self::ProcessedVerifyingKey, Self::Error> {
        Groth16::<E>::process_vk(vk).map_err(Mi355xError::from)
    }

    /// snark/src/lib.rs:76-80 -- unchanged CPU verifier; proofs are byte-compatible.
    fn verify_with_processed_vk(
        pvk: &Self::ProcessedVerifyingKey,
        public_input: &[E::ScalarField],
        proof: &Self::Proof,
    ) -> Result<bool, Self::Error> {
        Groth16::<E>::verify_with_processed_vk(pvk, public_input, proof).map_err(Mi355xError::from)
    }
}

/// snark/src/lib.rs:84-93: `setup` delegates to `circuit_specific_setup` (the trait's default body).
impl<E, P1, P2> CircuitSpecificSetupSNARK<E::ScalarField> for Mi355xGroth16<E>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
}
