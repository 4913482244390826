//! Device residency: one `ark355_ctx` per proving thread and the (proving key, R1CS) handles that live in HBM.
//!
//! `ConstraintSystemRef` is `Rc<RefCell<..>>` (relations/src/gr1cs/constraint_system_ref.rs:33), so the reference's
//! parallel unit is one OS thread per proof; accordingly every thread owns its context (the library serialises the
//! calls of one context internally).  Key handles are shared between threads: they are immutable after load.
//! The cache key is a fingerprint of the verifying key's serialized bytes plus the query lengths -- two different
//! proving keys never share it, and a dropped key's ~15 GB of window tables are released with `evict`.
use std::{
    cell::RefCell,
    collections::HashMap,
    sync::{Arc, Mutex, OnceLock},
};

use ark_ec::short_weierstrass::{Affine, SWCurveConfig};
use ark_ff::{Field, PrimeField};
use ark_groth16::ProvingKey;
use ark_relations::gr1cs::Matrix;
use ark_serialize::CanonicalSerialize;

use crate::{
    ffi,
    marshal::{csr_from_matrix, flatten_key, scalars_image, Mi355xCurve},
    Mi355xError,
};

/// Owning wrapper of an `ark355_ctx*`.
pub struct Ctx(pub *mut ffi::ark355_ctx);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { ffi::ark355_ctx_destroy(self.0) }
    }
}

thread_local! {
    static CTX: RefCell<Option<(i32, Ctx)>> = const { RefCell::new(None) };
}

/// GPU this thread proves on (default: `ARK355_DEVICE` or 0).  Call before the first proof of the thread.
pub fn set_device(device_id: i32) {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if c.as_ref().map(|(d, _)| *d) != Some(device_id) {
            *c = None;
        }
        DEVICE.with(|d| *d.borrow_mut() = Some(device_id));
    });
}
thread_local! {
    static DEVICE: RefCell<Option<i32>> = const { RefCell::new(None) };
}

pub fn last_error(ctx: *const ffi::ark355_ctx) -> String {
    unsafe {
        let p = ffi::ark355_last_error(ctx);
        if p.is_null() {
            String::new()
        } else {
            core::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

pub fn check(ctx: *const ffi::ark355_ctx, rc: i32) -> Result<(), Mi355xError> {
    if rc == ffi::ARK355_OK {
        Ok(())
    } else {
        Err(Mi355xError::from_code(rc, last_error(ctx)))
    }
}

/// Run `f` with this thread's context (created on first use).
pub fn with_ctx<T>(f: impl FnOnce(*mut ffi::ark355_ctx) -> Result<T, Mi355xError>) -> Result<T, Mi355xError> {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if c.is_none() {
            let dev = DEVICE
                .with(|d| *d.borrow())
                .or_else(|| std::env::var("ARK355_DEVICE").ok().and_then(|v| v.parse().ok()))
                .unwrap_or(0);
            let mut raw = core::ptr::null_mut();
            let rc = unsafe { ffi::ark355_ctx_create(dev, &mut raw) };
            if rc != ffi::ARK355_OK {
                return Err(Mi355xError::from_code(rc, "ark355_ctx_create".into()));
            }
            let ctx = Ctx(raw);
            if !trust_cache() {
                // every proof of this thread also checks a_i b_i = c_i on the device (include/ark355.h, policy CHECK_SATISFIED):
                // what makes the witness-only fast path safe and the outcome independent of the cache's state (lib.rs `prepare`)
                let rc = unsafe { ffi::ark355_ctx_set_policy(ctx.0, b"CHECK_SATISFIED\0".as_ptr() as *const core::ffi::c_char, 1) };
                if rc != ffi::ARK355_OK {
                    return Err(Mi355xError::from_code(rc, "ark355_ctx_set_policy(CHECK_SATISFIED)".into()));
                }
            }
            *c = Some((dev, ctx));
        }
        f(c.as_ref().unwrap().1 .0)
    })
}

/// `ARK_MI355X_TRUST_CACHE=1`: no satisfaction check on the device -- a proof is computed for whatever assignment the
/// circuit yields, as `ark-groth16`'s release build does.  For hosts that prove ONE circuit per key: with it, a circuit type
/// that builds different systems from call to call silently gets proofs over the wrong matrices.
pub fn trust_cache() -> bool {
    static TRUST: OnceLock<bool> = OnceLock::new();
    *TRUST.get_or_init(|| std::env::var("ARK_MI355X_TRUST_CACHE").map(|v| v == "1").unwrap_or(false))
}

/// Resident key + matrices of one circuit.
pub struct Resident {
    pub pk: *mut ffi::ark355_pk,
    pub r1cs: *mut ffi::ark355_r1cs,
    pub num_instance: usize,
    pub num_witness: usize,
    pub num_constraints: usize,
    /// Hash of the three matrices this entry was loaded with (`matrices_hash`): what tells two circuits apart whose proving
    /// keys share a sampled fingerprint.
    pub matrices_hash: MatHash,
    /// Circuit types (`core::any::type_name`) whose matrices were compared with this entry and found equal: only those
    /// may take the witness-only path.
    confirmed: Mutex<Vec<&'static str>>,
}
// SAFETY: the handles are immutable after load and the library allows concurrent readers (include/ark355.h).
unsafe impl Send for Resident {}
unsafe impl Sync for Resident {}
/// How a resident key sits in HBM (`ark355_pk_table_info`): Pippenger window size, windows per scalar, window stride of the
/// tables (1 = a table per window; larger = the key did not fit with full tables) and the bytes its five tables occupy.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct TableInfo {
    pub window_bits: u32,
    pub windows: u32,
    pub table_stride: u32,
    pub table_bytes: u64,
}
impl Resident {
    pub fn is_confirmed(&self, circuit_type: &'static str) -> bool {
        self.confirmed.lock().unwrap().iter().any(|t| *t == circuit_type)
    }
    pub fn confirm(&self, circuit_type: &'static str) {
        let mut c = self.confirmed.lock().unwrap();
        if !c.iter().any(|t| *t == circuit_type) {
            c.push(circuit_type);
        }
    }
    pub fn unconfirm(&self, circuit_type: &'static str) {
        self.confirmed.lock().unwrap().retain(|t| *t != circuit_type);
    }
    pub fn table_info(&self) -> TableInfo {
        let (mut c, mut w, mut s, mut b) = (0u32, 0u32, 0u32, 0u64);
        // SAFETY: `self.pk` is a live handle until drop; the out-pointers are valid for the call.
        unsafe { ffi::ark355_pk_table_info(self.pk, &mut c, &mut w, &mut s, &mut b) };
        TableInfo { window_bits: c, windows: w, table_stride: s, table_bytes: b }
    }
}
impl Drop for Resident {
    fn drop(&mut self) {
        unsafe {
            ffi::ark355_pk_free(self.pk);
            ffi::ark355_r1cs_free(self.r1cs);
        }
    }
}

type Key = [u8; 32];
pub type MatHash = [u8; 32];

/// Resident keys of this process, most recently used last.  BOUNDED: a key's window tables are ~25x its size (15 GB at
/// n = 2^20), so an unbounded map turns "one `pk.clone()` per request" into an out-of-memory condition.  When a load would
/// exceed `max_resident_keys()` the least recently used entry is dropped from the registry; its HBM is released as soon
/// as the last proof holding its `Arc` finishes.
struct Registry {
    entries: Vec<(Key, Arc<Resident>)>,
    /// (fingerprint, circuit type) pairs that lost the witness-only fast path: the library refused an assignment the type
    /// produced in witness-only mode, so either that circuit was unsatisfied or the type builds more than one system.
    /// Bounded like the entries (oldest dropped first).
    unstable: Vec<(Key, &'static str)>,
}
impl Registry {
    /// Every entry under this fingerprint (several circuits may share one, see `fingerprint`), most recently used last.
    fn get_all(&mut self, k: &Key) -> Vec<Arc<Resident>> {
        self.entries.iter().filter(|(key, _)| key == k).map(|(_, r)| r.clone()).collect()
    }
    fn touch(&mut self, k: &Key, r: &Arc<Resident>) {
        if let Some(i) = self.entries.iter().position(|(key, e)| key == k && Arc::ptr_eq(e, r)) {
            let e = self.entries.remove(i);
            self.entries.push(e);
        }
    }
    fn insert(&mut self, k: Key, r: Arc<Resident>) {
        self.entries.retain(|(key, e)| !(key == &k && e.matrices_hash == r.matrices_hash));
        while self.entries.len() >= max_resident_keys() {
            self.entries.remove(0);
        }
        self.entries.push((k, r));
    }
    fn remove(&mut self, k: &Key) {
        self.entries.retain(|(key, _)| key != k);
        self.unstable.retain(|(key, _)| key != k);
    }
}
/// `ARK_MI355X_MAX_RESIDENT_KEYS` (default 4; at least 1).
pub fn max_resident_keys() -> usize {
    static N: OnceLock<usize> = OnceLock::new();
    *N.get_or_init(|| std::env::var("ARK_MI355X_MAX_RESIDENT_KEYS").ok().and_then(|v| v.parse().ok()).filter(|n| *n >= 1).unwrap_or(4))
}
fn registry() -> &'static Mutex<Registry> {
    static R: OnceLock<Mutex<Registry>> = OnceLock::new();
    R.get_or_init(|| Mutex::new(Registry { entries: Vec::new(), unstable: Vec::new() }))
}

/// Fingerprint of a proving key, by CONTENT only: the verifying key, the two prover-only points, the query lengths and
/// 1024 evenly spaced elements (plus the last one) of each of the five query vectors.  Hashing all ~600 MB of a 2^20 key on
/// every `prove` is not an option, and the address of the key's buffers is deliberately NOT part of it (a clone or a move
/// of the same key must hit the cache instead of uploading another 10 GB of tables; a freed key whose buffer is reused by
/// a same-shape key must not alias).
///
/// The fingerprint is a PRE-FILTER, not an identity.  Two circuits of the same shape set up from the same deterministic
/// seed (the usual `test_rng` pattern) share `vk` (it depends on the instance columns only), `h_query` (it depends on the
/// domain only) and every length; their `a / b / l_query` differ only at the variables of the constraints that differ -- a
/// handful of elements out of millions, which 1024 samples miss with probability ~99.9 %.  What identifies a cache entry
/// is therefore the fingerprint AND the hash of the matrices it was loaded with (`matrices_hash`); several entries may
/// share a fingerprint.  The witness-only fast path (`Mi355xGroth16::prepare`) is open only to circuit TYPES whose matrices
/// were compared with the entry once, and every proof -- fast path or not -- has its assignment checked against the
/// matrices on the device as part of the witness map (policy `CHECK_SATISFIED`, set in `with_ctx`).
pub fn fingerprint<E: Mi355xCurve>(pk: &ProvingKey<E>) -> Key {
    let mut bytes = Vec::new();
    pk.vk.serialize_compressed(&mut bytes).expect("vk serialization");
    pk.beta_g1.serialize_compressed(&mut bytes).expect("point serialization");
    pk.delta_g1.serialize_compressed(&mut bytes).expect("point serialization");
    for n in [pk.a_query.len(), pk.b_g1_query.len(), pk.b_g2_query.len(), pk.h_query.len(), pk.l_query.len(), E::CURVE_ID as usize] {
        bytes.extend_from_slice(&(n as u64).to_le_bytes());
    }
    fn sample<T: CanonicalSerialize>(v: &[T], bytes: &mut Vec<u8>) {
        if v.is_empty() {
            return;
        }
        let step = core::cmp::max(1, v.len() / 1024);
        let mut i = 0;
        while i < v.len() {
            v[i].serialize_uncompressed(&mut *bytes).expect("point serialization");
            i += step;
        }
        v[v.len() - 1].serialize_uncompressed(&mut *bytes).expect("point serialization");
    }
    sample(&pk.a_query, &mut bytes);
    sample(&pk.b_g1_query, &mut bytes);
    sample(&pk.b_g2_query, &mut bytes);
    sample(&pk.h_query, &mut bytes);
    sample(&pk.l_query, &mut bytes);
    let mut lanes = [0xcbf29ce484222325u64, 0x84222325cbf29ce4, 0x9e3779b97f4a7c15, 0xd6e8feb86659fd93];
    for (i, b) in bytes.iter().enumerate() {
        let l = &mut lanes[i & 3];
        *l = (*l ^ (*b as u64)).wrapping_mul(0x100000001b3);
        *l ^= *l >> 29;
    }
    let mut out = [0u8; 32];
    for (i, l) in lanes.iter().enumerate() {
        out[8 * i..8 * i + 8].copy_from_slice(&l.to_le_bytes());
    }
    out
}

/// The most recently used entry under this key's fingerprint (callers that hold ONE circuit per key: `prove_assignments`,
/// the pool); the entry is marked as used.  `SNARK::prove` goes through `candidates` and confirms the entry against its circuit.
pub fn lookup<E: Mi355xCurve>(pk: &ProvingKey<E>) -> Option<Arc<Resident>> {
    let k = fingerprint(pk);
    let mut reg = registry().lock().unwrap();
    let hit = reg.get_all(&k).pop();
    if let Some(r) = &hit {
        reg.touch(&k, r);
    }
    hit
}

/// Has `circuit_type` lost the witness-only fast path for keys with this fingerprint (`mark_unstable`)?
pub fn is_unstable<E: Mi355xCurve>(pk: &ProvingKey<E>, circuit_type: &'static str) -> bool {
    let k = fingerprint(pk);
    registry().lock().unwrap().unstable.iter().any(|(key, t)| key == &k && *t == circuit_type)
}

/// `circuit_type` takes the full path (matrices rebuilt and compared) for keys with this fingerprint from now on.
pub fn mark_unstable<E: Mi355xCurve>(pk: &ProvingKey<E>, circuit_type: &'static str) {
    let k = fingerprint(pk);
    let mut reg = registry().lock().unwrap();
    if !reg.unstable.iter().any(|(key, t)| key == &k && *t == circuit_type) {
        if reg.unstable.len() >= 64 {
            reg.unstable.remove(0);
        }
        reg.unstable.push((k, circuit_type));
    }
}

/// Every entry whose key has this key's fingerprint.
pub fn candidates<E: Mi355xCurve>(pk: &ProvingKey<E>) -> Vec<Arc<Resident>> {
    registry().lock().unwrap().get_all(&fingerprint(pk))
}

/// Mark `res` as the most recently used entry of its fingerprint.
pub fn touch<E: Mi355xCurve>(pk: &ProvingKey<E>, res: &Arc<Resident>) {
    registry().lock().unwrap().touch(&fingerprint(pk), res);
}

/// Hash of the R1CS matrices (rows, columns, canonical coefficients) and the constraint count: what tells two cache entries
/// under one fingerprint apart.  Not a cryptographic hash (the inputs are the caller's own circuits, not an adversary's), but
/// every word reaches all four 64-bit lanes -- each lane absorbs the word under its own odd multiplier and is stirred by a
/// splitmix64 finaliser -- so the 256-bit value does not degenerate into four independent hashes of a quarter of the data.
pub fn matrices_hash<F: PrimeField>(matrices: &[Matrix<F>], num_constraints: usize) -> MatHash {
    const MUL: [u64; 4] = [0x9e3779b97f4a7c15, 0xbf58476d1ce4e5b9, 0x94d049bb133111eb, 0xd6e8feb86659fd93];
    let mut lanes = [0xcbf29ce484222325u64, 0x84222325cbf29ce4, 0x9e3779b97f4a7c15, 0xd6e8feb86659fd93];
    let mut feed = |w: u64| {
        for (l, m) in lanes.iter_mut().zip(MUL.iter()) {
            let mut x = (*l ^ w).wrapping_mul(*m);
            x ^= x >> 30;
            x = x.wrapping_mul(0xbf58476d1ce4e5b9);
            x ^= x >> 27;
            *l = x.rotate_left(23).wrapping_add(w);
        }
    };
    feed(num_constraints as u64);
    feed(matrices.len() as u64);
    for m in matrices {
        feed(m.len() as u64);
        for row in m {
            feed(row.len() as u64);
            for (coeff, col) in row {
                feed(*col as u64);
                for limb in coeff.into_bigint().as_ref() {
                    feed(*limb);
                }
            }
        }
    }
    let mut out = [0u8; 32];
    for (i, l) in lanes.iter().enumerate() {
        out[8 * i..8 * i + 8].copy_from_slice(&l.to_le_bytes());
    }
    out
}

/// Release the HBM of a key (window tables: ~15 GB at n = 2^20) once no proof uses it any more.
pub fn evict<E: Mi355xCurve>(pk: &ProvingKey<E>) {
    registry().lock().unwrap().remove(&fingerprint(pk));
}

/// First proof of a circuit: upload the key (the library builds its per-window tables in HBM) and the three R1CS
/// matrices of `ConstraintSystem::to_matrices()["R1CS"]` (relations/src/gr1cs/constraint_system.rs:768-774).
pub fn load<E, P1, P2>(
    pk: &ProvingKey<E>,
    matrices: &[Matrix<E::ScalarField>],
    num_constraints: usize,
    num_instance: usize,
    num_witness: usize,
) -> Result<Arc<Resident>, Mi355xError>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    if matrices.len() != 3 {
        return Err(Mi355xError::InvalidArgument("the R1CS predicate has three matrices".into()));
    }
    let flat = flatten_key::<E, P1, P2>(pk)?;
    if flat.num_instance as usize != num_instance || flat.num_witness as usize != num_witness {
        return Err(Mi355xError::InvalidArgument("proving key and constraint system dimensions differ".into()));
    }
    let csr: Vec<_> = matrices.iter().map(csr_from_matrix).collect();
    let res = with_ctx(|ctx| {
        let desc = flat.desc();
        let mut pk_h = core::ptr::null_mut();
        check(ctx, unsafe { ffi::ark355_pk_load(ctx, E::CURVE_ID, &desc, &mut pk_h) })?;
        let row_ptr = [csr[0].row_ptr.as_ptr(), csr[1].row_ptr.as_ptr(), csr[2].row_ptr.as_ptr()];
        let col = [csr[0].col.as_ptr(), csr[1].col.as_ptr(), csr[2].col.as_ptr()];
        let coeff =
            [scalars_image(&csr[0].coeff).as_ptr(), scalars_image(&csr[1].coeff).as_ptr(), scalars_image(&csr[2].coeff).as_ptr()];
        let mut r1_h = core::ptr::null_mut();
        let rc = unsafe {
            ffi::ark355_r1cs_load(
                ctx,
                E::CURVE_ID,
                num_constraints as u64,
                num_instance as u64,
                num_witness as u64,
                row_ptr.as_ptr(),
                col.as_ptr(),
                coeff.as_ptr(),
                &mut r1_h,
            )
        };
        if rc != ffi::ARK355_OK {
            let e = Mi355xError::from_code(rc, last_error(ctx));
            unsafe { ffi::ark355_pk_free(pk_h) };
            return Err(e);
        }
        Ok(Resident {
            pk: pk_h,
            r1cs: r1_h,
            num_instance,
            num_witness,
            num_constraints,
            matrices_hash: matrices_hash(matrices, num_constraints),
            confirmed: Mutex::new(Vec::new()),
        })
    })?;
    let res = Arc::new(res);
    registry().lock().unwrap().insert(fingerprint(pk), res.clone());
    Ok(res)
}
