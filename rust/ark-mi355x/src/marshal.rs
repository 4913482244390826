//! Memory images handed across the C ABI (include/ark355.h, "Conventions").
//!
//! * `Fr`, `Fq`: ark-ff's `Fp<MontBackend<_, N>, N>` is `Fp(pub BigInt<N>, PhantomData)`, i.e. `[u64; N]` little-endian
//!   limbs in MONTGOMERY form -- exactly what the library consumes, so `&[Fr]` crosses as `*const u8` without conversion.
//!   `layout_self_test` checks that assumption at start-up instead of trusting it.
//! * points: `x || y` (G2: `x.c0 || x.c1 || y.c0 || y.c1`), infinity = all zero bytes.  `Affine {x, y, infinity}` is
//!   `repr(Rust)`, so coordinates are copied explicitly.
//! * `r`, `s` and MSM scalars: canonical (`into_bigint`) 32-byte little-endian integers.
use ark_ec::{pairing::Pairing, short_weierstrass::{Affine, SWCurveConfig}, AffineRepr};
use ark_ff::{BigInteger, Field, PrimeField, Zero};
use ark_groth16::{Proof, ProvingKey};
use ark_relations::gr1cs::Matrix;

use crate::{ffi, Mi355xError};

/// Pairing engines the library is instantiated for.
pub trait Mi355xCurve: Pairing {
    /// `ARK355_BLS12_381` / `ARK355_BN254`
    const CURVE_ID: i32;
    /// bytes of one base-field element image
    const FQ_BYTES: usize;
}
impl Mi355xCurve for ark_bls12_381::Bls12_381 {
    const CURVE_ID: i32 = ffi::ARK355_BLS12_381;
    const FQ_BYTES: usize = 48;
}
impl Mi355xCurve for ark_bn254::Bn254 {
    const CURVE_ID: i32 = ffi::ARK355_BN254;
    const FQ_BYTES: usize = 32;
}

/// Montgomery image of a prime-field element (see the module docs).
#[inline]
fn prime_image<F: PrimeField>(f: &F) -> &[u8] {
    // SAFETY: F = Fp<MontBackend<..>, N> is a newtype over BigInt<N> = [u64; N] plus a zero-sized marker.
    unsafe { core::slice::from_raw_parts(f as *const F as *const u8, core::mem::size_of::<F>()) }
}

#[inline]
fn prime_from_image<F: PrimeField>(b: &[u8]) -> F {
    assert_eq!(b.len(), core::mem::size_of::<F>());
    // SAFETY: every bit pattern the library returns is a reduced Montgomery image; same layout argument as above.
    unsafe { core::ptr::read_unaligned(b.as_ptr() as *const F) }
}

/// Image of an element of the base field of G1 or G2: its prime-field components in order (c0, c1).
fn push_field_image<F: Field>(f: &F, out: &mut Vec<u8>)
where
    F::BasePrimeField: PrimeField,
{
    for c in f.to_base_prime_field_elements() {
        out.extend_from_slice(prime_image(&c));
    }
}

fn field_from_image<F: Field>(b: &[u8]) -> F
where
    F::BasePrimeField: PrimeField,
{
    let k = core::mem::size_of::<F::BasePrimeField>();
    let comps: Vec<F::BasePrimeField> = b.chunks_exact(k).map(prime_from_image::<F::BasePrimeField>).collect();
    F::from_base_prime_field_elems(comps).expect("component count")
}

/// `x || y`, or zeros for the point at infinity.
pub fn push_point<P: SWCurveConfig>(p: &Affine<P>, out: &mut Vec<u8>)
where
    <P::BaseField as Field>::BasePrimeField: PrimeField,
{
    let coord = core::mem::size_of::<<P::BaseField as Field>::BasePrimeField>() * P::BaseField::extension_degree() as usize;
    match p.xy() {
        Some((x, y)) => {
            push_field_image(&x, out);
            push_field_image(&y, out);
        },
        None => out.resize(out.len() + 2 * coord, 0),
    }
}

pub fn point_from_image<P: SWCurveConfig>(b: &[u8]) -> Affine<P>
where
    <P::BaseField as Field>::BasePrimeField: PrimeField,
{
    if b.iter().all(|v| *v == 0) {
        return Affine::<P>::identity();
    }
    let half = b.len() / 2;
    // the library only returns points it computed: already on the curve and in the subgroup
    Affine::<P>::new_unchecked(field_from_image(&b[..half]), field_from_image(&b[half..]))
}

pub fn flatten_points<P: SWCurveConfig>(v: &[Affine<P>]) -> Vec<u8>
where
    <P::BaseField as Field>::BasePrimeField: PrimeField,
{
    let mut out = Vec::new();
    for p in v {
        push_point(p, &mut out);
    }
    out
}

/// Scalars as one contiguous Montgomery image: the assignment z goes across without any conversion.
pub fn scalars_image<F: PrimeField>(z: &[F]) -> &[u8] {
    // SAFETY: see `prime_image`; a slice of Fp is a contiguous array of [u64; N].
    unsafe { core::slice::from_raw_parts(z.as_ptr() as *const u8, core::mem::size_of_val(z)) }
}

pub fn canonical_32<F: PrimeField>(f: &F) -> [u8; 32] {
    let mut out = [0u8; 32];
    let b = f.into_bigint().to_bytes_le();
    out[..b.len()].copy_from_slice(&b);
    out
}

/// `Matrix<F> = Vec<Vec<(F, usize)>>` (relations/src/utils/matrix.rs:4) -> CSR (row_ptr u64, col u32, coeff images).
pub struct Csr<F: PrimeField> {
    pub row_ptr: Vec<u64>,
    pub col: Vec<u32>,
    pub coeff: Vec<F>,
}

pub fn csr_from_matrix<F: PrimeField>(m: &Matrix<F>) -> Csr<F> {
    let nnz: usize = m.iter().map(|r| r.len()).sum();
    let mut out = Csr { row_ptr: Vec::with_capacity(m.len() + 1), col: Vec::with_capacity(nnz), coeff: Vec::with_capacity(nnz) };
    out.row_ptr.push(0);
    for row in m {
        for (c, j) in row {
            out.col.push(*j as u32);
            out.coeff.push(*c);
        }
        out.row_ptr.push(out.col.len() as u64);
    }
    out
}

/// The five query vectors and five single points of `ark_groth16::ProvingKey<E>` as the images `ark355_pk_desc` points at.
pub struct FlatKey {
    pub a_query: Vec<u8>,
    pub b_g1_query: Vec<u8>,
    pub b_g2_query: Vec<u8>,
    pub h_query: Vec<u8>,
    pub l_query: Vec<u8>,
    pub alpha_g1: Vec<u8>,
    pub beta_g1: Vec<u8>,
    pub delta_g1: Vec<u8>,
    pub beta_g2: Vec<u8>,
    pub delta_g2: Vec<u8>,
    pub num_instance: u64,
    pub num_witness: u64,
    pub domain_size: u64,
}

pub fn flatten_key<E, P1, P2>(pk: &ProvingKey<E>) -> Result<FlatKey, Mi355xError>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    let m = pk.a_query.len();
    let w = pk.l_query.len();
    if pk.b_g1_query.len() != m || pk.b_g2_query.len() != m || w > m || pk.vk.gamma_abc_g1.len() != m - w {
        return Err(Mi355xError::InvalidArgument("proving key: inconsistent query lengths".into()));
    }
    let n_dom = pk.h_query.len() + 1;
    if !n_dom.is_power_of_two() {
        return Err(Mi355xError::InvalidArgument("proving key: h_query length + 1 is not a power of two".into()));
    }
    let one = |p: &Affine<P1>| flatten_points(core::slice::from_ref(p));
    let one2 = |p: &Affine<P2>| flatten_points(core::slice::from_ref(p));
    Ok(FlatKey {
        a_query: flatten_points(&pk.a_query),
        b_g1_query: flatten_points(&pk.b_g1_query),
        b_g2_query: flatten_points(&pk.b_g2_query),
        h_query: flatten_points(&pk.h_query),
        l_query: flatten_points(&pk.l_query),
        alpha_g1: one(&pk.vk.alpha_g1),
        beta_g1: one(&pk.beta_g1),
        delta_g1: one(&pk.delta_g1),
        beta_g2: one2(&pk.vk.beta_g2),
        delta_g2: one2(&pk.vk.delta_g2),
        num_instance: (m - w) as u64,
        num_witness: w as u64,
        domain_size: n_dom as u64,
    })
}

impl FlatKey {
    pub fn desc(&self) -> ffi::ark355_pk_desc {
        ffi::ark355_pk_desc {
            num_instance: self.num_instance,
            num_witness: self.num_witness,
            domain_size: self.domain_size,
            a_query: self.a_query.as_ptr(),
            b_g1_query: self.b_g1_query.as_ptr(),
            b_g2_query: self.b_g2_query.as_ptr(),
            h_query: self.h_query.as_ptr(),
            l_query: self.l_query.as_ptr(),
            alpha_g1: self.alpha_g1.as_ptr(),
            beta_g1: self.beta_g1.as_ptr(),
            delta_g1: self.delta_g1.as_ptr(),
            beta_g2: self.beta_g2.as_ptr(),
            delta_g2: self.delta_g2.as_ptr(),
        }
    }
}

/// `ark355_proof_raw` -> `ark_groth16::Proof<E>`: rebuild `Affine {x, y, infinity}` from the images.
pub fn proof_from_raw<E, P1, P2>(raw: &ffi::ark355_proof_raw) -> Proof<E>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    let (g1, g2) = (2 * E::FQ_BYTES, 4 * E::FQ_BYTES);
    Proof { a: point_from_image::<P1>(&raw.a[..g1]), b: point_from_image::<P2>(&raw.b[..g2]), c: point_from_image::<P1>(&raw.c[..g1]) }
}

/// Start-up check of the layout assumptions above: the image of ONE is R mod p and survives a round trip; sizes
/// agree with `ark355_sizes`.
pub fn layout_self_test<E: Mi355xCurve>() -> Result<(), Mi355xError>
where
    <E::G1Affine as AffineRepr>::BaseField: PrimeField,
{
    let mut sizes = [0u32; 4];
    let rc = unsafe { ffi::ark355_sizes(E::CURVE_ID, sizes.as_mut_ptr()) };
    if rc != ffi::ARK355_OK {
        return Err(Mi355xError::from_code(rc, "ark355_sizes".into()));
    }
    let fr = core::mem::size_of::<E::ScalarField>();
    let fq = core::mem::size_of::<<E::G1Affine as AffineRepr>::BaseField>();
    if sizes[0] as usize != fr || sizes[1] as usize != fq || fq != E::FQ_BYTES {
        return Err(Mi355xError::InvalidArgument(format!("field sizes differ: library {sizes:?}, ark-ff Fr {fr} Fq {fq}")));
    }
    let one = E::ScalarField::from(1u64);
    let two = E::ScalarField::from(2u64);
    if prime_from_image::<E::ScalarField>(prime_image(&one)) != one || prime_image(&one) == prime_image(&two) || one.is_zero() {
        return Err(Mi355xError::InvalidArgument("ark-ff field layout is not the expected Montgomery limb image".into()));
    }
    // Montgomery, not canonical: the image of 1 must not be the integer 1 (R mod p != 1 for these fields)
    let img = prime_image(&one);
    if img[0] == 1 && img[1..].iter().all(|b| *b == 0) {
        return Err(Mi355xError::InvalidArgument("ark-ff field elements are not stored in Montgomery form".into()));
    }
    Ok(())
}
