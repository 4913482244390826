//! The resident-key cache of the shim (src/cache.rs, src/lib.rs `prepare`) on a machine with both toolchains: what a caller
//! observes must not depend on what the cache holds.
//!   * two circuits of one shape whose keys come from the same deterministic seed share the SAMPLED key fingerprint: both must
//!     get their own cache entry (matrices hash) and proofs that verify;
//!   * an unsatisfied witness is refused with `SynthesisError::Unsatisfiable` on the FIRST call of a key (full path) and on
//!     the SECOND (witness-only fast path) alike;
//!   * a circuit TYPE that builds two different systems loses the fast path after at most one refusal and proves both
//!     systems correctly afterwards.
use ark_bls12_381::{Bls12_381, Fr};
use ark_groth16::Groth16;
use ark_mi355x::{Mi355xError, Mi355xGroth16};
use ark_relations::gr1cs::{ConstraintSynthesizer, ConstraintSystemRef, LinearCombination, SynthesisError, Variable};
use ark_snark::SNARK;
use ark_std::rand::{rngs::StdRng, SeedableRng};

/// `len` multiplications x_{i+1} = x_i * (x_i or w) -- `twist` selects which, at ONE constraint in the middle: two systems of
/// the same shape (same counts of everything) that differ in a single row of B.
#[derive(Clone)]
struct Chain {
    len: usize,
    twist: bool,
    x0: Fr,
    w: Fr,
    /// added to the claimed public output: non-zero makes the assignment unsatisfied (last constraint)
    lie: Fr,
}

impl ConstraintSynthesizer<Fr> for Chain {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> Result<(), SynthesisError> {
        let mut vals = vec![self.x0];
        for i in 0..self.len {
            let x = vals[i];
            let rhs = if self.twist && i == self.len / 2 { self.w } else { x };
            vals.push(x * rhs);
        }
        let out = cs.new_input_variable(|| Ok(vals[self.len] + self.lie))?;
        let w = cs.new_witness_variable(|| Ok(self.w))?;
        let mut vars = Vec::with_capacity(self.len + 1);
        for v in &vals {
            vars.push(cs.new_witness_variable(|| Ok(*v))?);
        }
        for i in 0..self.len {
            let rhs = if self.twist && i == self.len / 2 { w } else { vars[i] };
            cs.enforce_r1cs_constraint(|| LinearCombination::from(vars[i]), || LinearCombination::from(rhs), || LinearCombination::from(vars[i + 1]))?;
        }
        cs.enforce_r1cs_constraint(|| LinearCombination::from(vars[self.len]), || LinearCombination::from(Variable::One), || LinearCombination::from(out))?;
        Ok(())
    }
}

fn chain(twist: bool, lie: u64) -> Chain {
    Chain { len: 4096, twist, x0: Fr::from(3u64), w: Fr::from(7u64), lie: Fr::from(lie) }
}

fn public(c: &Chain) -> Vec<Fr> {
    let mut x = c.x0;
    for i in 0..c.len {
        x *= if c.twist && i == c.len / 2 { c.w } else { x };
    }
    vec![x + c.lie]
}

fn unsatisfiable<T: core::fmt::Debug>(r: Result<T, Mi355xError>) -> bool {
    matches!(r, Err(Mi355xError::Synthesis(SynthesisError::Unsatisfiable)))
}

#[test]
fn same_seed_keys_of_two_systems_get_their_own_entries() {
    let (plain, twisted) = (chain(false, 0), chain(true, 0));
    // the usual test pattern: every setup from the same seed => same toxic waste => vk, h_query and all lengths coincide
    let (pk_a, vk_a) = Groth16::<Bls12_381>::circuit_specific_setup(plain.clone(), &mut StdRng::seed_from_u64(1)).unwrap();
    let (pk_b, vk_b) = Groth16::<Bls12_381>::circuit_specific_setup(twisted.clone(), &mut StdRng::seed_from_u64(1)).unwrap();
    let mut rng = StdRng::seed_from_u64(2);
    for round in 0..3 {
        let pa = Mi355xGroth16::<Bls12_381>::prove(&pk_a, plain.clone(), &mut rng).unwrap();
        assert!(Mi355xGroth16::<Bls12_381>::verify(&vk_a, &public(&plain), &pa).unwrap(), "plain, round {round}");
        // one circuit TYPE, two systems: the second key's first proof may be refused once (fast path against the first key's
        // matrices), never more than once, and every accepted proof verifies
        let mut refused = 0;
        let pb = loop {
            match Mi355xGroth16::<Bls12_381>::prove(&pk_b, twisted.clone(), &mut rng) {
                Ok(p) => break p,
                Err(e) => {
                    assert!(unsatisfiable::<()>(Err(e)));
                    refused += 1;
                    assert!(refused <= 1 && round == 0, "a type that builds two systems is refused at most once per key fingerprint");
                }
            }
        };
        assert!(Mi355xGroth16::<Bls12_381>::verify(&vk_b, &public(&twisted), &pb).unwrap(), "twisted, round {round}");
    }
    ark_mi355x::evict(&pk_a);
    ark_mi355x::evict(&pk_b);
}

#[test]
fn an_unsatisfied_witness_is_refused_on_the_first_call_and_on_cached_calls() {
    let good = chain(false, 0);
    let bad = chain(false, 1);
    let (pk, vk) = Groth16::<Bls12_381>::circuit_specific_setup(good.clone(), &mut StdRng::seed_from_u64(11)).unwrap();
    let mut rng = StdRng::seed_from_u64(12);
    // first call of the key: full path
    assert!(unsatisfiable(Mi355xGroth16::<Bls12_381>::prove(&pk, bad.clone(), &mut rng)));
    // a satisfied circuit of the same type proves (witness-only fast path: the first call loaded and confirmed the entry)
    let p = Mi355xGroth16::<Bls12_381>::prove(&pk, good.clone(), &mut rng).unwrap();
    assert!(Mi355xGroth16::<Bls12_381>::verify(&vk, &public(&good), &p).unwrap());
    // cached calls: refused the same way, and a satisfied circuit still proves afterwards
    for _ in 0..2 {
        assert!(unsatisfiable(Mi355xGroth16::<Bls12_381>::prove(&pk, bad.clone(), &mut rng)));
        let p = Mi355xGroth16::<Bls12_381>::prove(&pk, good.clone(), &mut rng).unwrap();
        assert!(Mi355xGroth16::<Bls12_381>::verify(&vk, &public(&good), &p).unwrap());
    }
    ark_mi355x::evict(&pk);
}

#[test]
fn matrices_hash_separates_a_single_changed_entry() {
    use ark_relations::gr1cs::{ConstraintSystem, R1CS_PREDICATE_LABEL};
    let mats = |c: Chain| {
        let cs = ConstraintSystem::<Fr>::new_ref();
        c.generate_constraints(cs.clone()).unwrap();
        cs.finalize();
        (cs.to_matrices().unwrap().remove(R1CS_PREDICATE_LABEL).unwrap(), cs.num_constraints())
    };
    let (a, na) = mats(chain(false, 0));
    let (b, nb) = mats(chain(true, 0));
    assert_eq!(na, nb);
    assert_ne!(ark_mi355x::matrices_hash(&a, na), ark_mi355x::matrices_hash(&b, nb));
    assert_eq!(ark_mi355x::matrices_hash(&a, na), ark_mi355x::matrices_hash(&mats(chain(false, 5)).0, na), "values are not structure");
}
