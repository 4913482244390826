//! End-to-end proving with synthesis in the loop: a `ProverPool` owns `inflight` long-lived prover
//! threads -- each with its own `ark355_ctx`, i.e. its own streams and gigabytes of scratch that are allocated once --
//! and proves assignments handed to it in page-locked buffers (`ark355_host_alloc`), while the caller's synthesis
//! threads keep producing the next ones.  Same arrangement as `Groth16::prove_pipelined` of the C++ mirror
//! (host_mirror/snark.hpp), which measured 38 M constraints/s end to end at 2^20 constraints against 43 M/s for
//! assignments synthesised up front.
//!
//! The reference's parallel unit is one OS thread per constraint system (`ConstraintSystemRef` is `Rc<RefCell<..>>`,
//! relations/src/gr1cs/constraint_system_ref.rs:33), so synthesis is parallel ACROSS proofs: `prove_pipelined` runs
//! `generate_constraints` for different instances on `synth_threads` scoped threads.
use std::{
    marker::PhantomData,
    sync::{
        mpsc::{channel, sync_channel, Receiver, Sender, SyncSender},
        Arc, Mutex,
    },
    thread::JoinHandle,
};

use ark_ec::short_weierstrass::{Affine, SWCurveConfig};
use ark_ff::{Field, PrimeField};
use ark_groth16::{Proof, ProvingKey};
use ark_relations::gr1cs::{ConstraintSynthesizer, SynthesisError};

use crate::{cache, ffi, marshal, Mi355xCurve, Mi355xError, Mi355xGroth16};

/// A full assignment `instance || witness` in page-locked host memory: the H2D copy inside `ark355_prove` runs at PCIe
/// rate instead of going through the runtime's bounce buffers.  Recycled by the pool that handed it out.
pub struct PinnedAssignment<F: PrimeField> {
    ptr: *mut F,
    len: usize,
    cap_bytes: usize,
}
// SAFETY: plain memory owned by the value.
unsafe impl<F: PrimeField> Send for PinnedAssignment<F> {}
impl<F: PrimeField> PinnedAssignment<F> {
    fn with_capacity(elements: usize) -> Result<Self, Mi355xError> {
        let bytes = elements * core::mem::size_of::<F>();
        let mut raw = core::ptr::null_mut();
        let rc = unsafe { ffi::ark355_host_alloc(bytes as u64, &mut raw) };
        if rc != ffi::ARK355_OK {
            return Err(Mi355xError::from_code(rc, "ark355_host_alloc".into()));
        }
        Ok(Self { ptr: raw as *mut F, len: 0, cap_bytes: bytes })
    }
    /// Copy `instance` then `witness` into the buffer (constraint_system.rs:193-206 order).
    pub fn fill(&mut self, instance: &[F], witness: &[F]) {
        let n = instance.len() + witness.len();
        assert!(n * core::mem::size_of::<F>() <= self.cap_bytes, "assignment longer than the buffer");
        unsafe {
            core::ptr::copy_nonoverlapping(instance.as_ptr(), self.ptr, instance.len());
            core::ptr::copy_nonoverlapping(witness.as_ptr(), self.ptr.add(instance.len()), witness.len());
        }
        self.len = n;
    }
    pub fn as_slice(&self) -> &[F] {
        unsafe { core::slice::from_raw_parts(self.ptr, self.len) }
    }
}
impl<F: PrimeField> Drop for PinnedAssignment<F> {
    fn drop(&mut self) {
        unsafe { ffi::ark355_host_free(self.ptr as *mut core::ffi::c_void) }
    }
}

struct Job<E: Mi355xCurve> {
    z: PinnedAssignment<E::ScalarField>,
    r: [u8; 32],
    s: [u8; 32],
    done: Sender<Result<Proof<E>, Mi355xError>>,
}

/// `inflight` prover threads over one resident key.  Dropping the pool joins the threads (and with them their device
/// contexts).
pub struct ProverPool<E: Mi355xCurve> {
    jobs: Option<SyncSender<Job<E>>>,
    workers: Vec<JoinHandle<()>>,
    free: Arc<Mutex<Vec<PinnedAssignment<E::ScalarField>>>>,
    res: Arc<cache::Resident>,
    _e: PhantomData<E>,
}

/// The proof of one submitted assignment.
pub struct Ticket<E: Mi355xCurve>(Receiver<Result<Proof<E>, Mi355xError>>);
impl<E: Mi355xCurve> Ticket<E> {
    pub fn wait(self) -> Result<Proof<E>, Mi355xError> {
        self.0.recv().unwrap_or_else(|_| Err(Mi355xError::Hip("prover thread terminated".into())))
    }
}

impl<E, P1, P2> ProverPool<E>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>> + 'static,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    /// `device`: the GPU the pool proves on; at most `2 * inflight` assignments wait in its queue.
    pub fn new(res: Arc<cache::Resident>, device: i32, inflight: usize) -> Self {
        let inflight = inflight.clamp(1, 16);
        let (tx, rx) = sync_channel::<Job<E>>(2 * inflight);
        let rx = Arc::new(Mutex::new(rx));
        let free = Arc::new(Mutex::new(Vec::new()));
        let mut workers = Vec::with_capacity(inflight);
        for _ in 0..inflight {
            let (rx, res, free) = (rx.clone(), res.clone(), free.clone());
            workers.push(std::thread::spawn(move || {
                cache::set_device(device); // this thread's context lives as long as the thread
                loop {
                    let job = match rx.lock().unwrap().recv() {
                        Ok(j) => j,
                        Err(_) => return, // pool dropped
                    };
                    let mut raw = ffi::ark355_proof_raw { a: [0; 96], b: [0; 192], c: [0; 96] };
                    let out = cache::with_ctx(|ctx| {
                        let zi = marshal::scalars_image(job.z.as_slice());
                        cache::check(ctx, unsafe {
                            ffi::ark355_prove(ctx, res.pk, res.r1cs, zi.as_ptr(), job.z.len as u64, job.r.as_ptr(), job.s.as_ptr(), &mut raw)
                        })
                    })
                    .map(|_| marshal::proof_from_raw::<E, P1, P2>(&raw));
                    free.lock().unwrap().push(job.z);
                    let _ = job.done.send(out);
                }
            }));
        }
        Self { jobs: Some(tx), workers, free, res, _e: PhantomData }
    }

    /// A page-locked buffer for one assignment of this key (recycled when its proof is done).
    pub fn buffer(&self) -> Result<PinnedAssignment<E::ScalarField>, Mi355xError> {
        if let Some(b) = self.free.lock().unwrap().pop() {
            return Ok(b);
        }
        PinnedAssignment::with_capacity(self.res.num_instance + self.res.num_witness)
    }

    /// Queue one proof (blocks while `2 * inflight` assignments are already waiting).
    pub fn submit(&self, z: PinnedAssignment<E::ScalarField>, r: E::ScalarField, s: E::ScalarField) -> Result<Ticket<E>, Mi355xError> {
        if z.len != self.res.num_instance + self.res.num_witness {
            return Err(Mi355xError::Synthesis(SynthesisError::AssignmentMissing));
        }
        let (done, ticket) = channel();
        let job = Job { z, r: marshal::canonical_32(&r), s: marshal::canonical_32(&s), done };
        self.jobs.as_ref().unwrap().send(job).map_err(|_| Mi355xError::Hip("prover pool is shut down".into()))?;
        Ok(Ticket(ticket))
    }
}

impl<E: Mi355xCurve> Drop for ProverPool<E> {
    fn drop(&mut self) {
        self.jobs = None; // closes the queue: the workers return from recv()
        for w in self.workers.drain(..) {
            let _ = w.join();
        }
    }
}

impl<E, P1, P2> Mi355xGroth16<E>
where
    E: Mi355xCurve<G1Affine = Affine<P1>, G2Affine = Affine<P2>> + 'static,
    P1: SWCurveConfig,
    P2: SWCurveConfig,
    <P1::BaseField as Field>::BasePrimeField: PrimeField,
    <P2::BaseField as Field>::BasePrimeField: PrimeField,
{
    /// `count` proofs of one circuit, synthesis overlapped with the GPU: instance `i` is built by `make_circuit(i)` and
    /// synthesised (witness-only mode) on one of `synth_threads` scoped threads, proved by `pool` with the randomisers
    /// `rs[i] = (r_i, s_i)`.  The key must be resident (prove once with `SNARK::prove`, then `cache::lookup`).
    pub fn prove_pipelined<C, M>(
        pool: &ProverPool<E>,
        count: usize,
        make_circuit: M,
        rs: &[(E::ScalarField, E::ScalarField)],
        synth_threads: usize,
    ) -> Result<Vec<Proof<E>>, Mi355xError>
    where
        C: ConstraintSynthesizer<E::ScalarField>,
        M: Fn(usize) -> C + Sync,
    {
        if rs.len() < count {
            return Err(Mi355xError::InvalidArgument("one (r, s) pair per proof".into()));
        }
        let next = std::sync::atomic::AtomicUsize::new(0);
        let tickets: Mutex<Vec<Option<Ticket<E>>>> = Mutex::new((0..count).map(|_| None).collect());
        let first_error: Mutex<Option<Mi355xError>> = Mutex::new(None);
        std::thread::scope(|scope| {
            for _ in 0..synth_threads.max(1) {
                scope.spawn(|| loop {
                    let i = next.fetch_add(1, std::sync::atomic::Ordering::Relaxed);
                    if i >= count || first_error.lock().unwrap().is_some() {
                        return;
                    }
                    let step = || -> Result<Ticket<E>, Mi355xError> {
                        let syn = crate::synthesize(make_circuit(i), true)?;
                        let mut buf = pool.buffer()?;
                        buf.fill(&syn.z, &[]);
                        pool.submit(buf, rs[i].0, rs[i].1)
                    };
                    match step() {
                        Ok(t) => tickets.lock().unwrap()[i] = Some(t),
                        Err(e) => {
                            first_error.lock().unwrap().get_or_insert(e);
                            return;
                        },
                    }
                });
            }
        });
        if let Some(e) = first_error.into_inner().unwrap() {
            return Err(e);
        }
        tickets.into_inner().unwrap().into_iter().map(|t| t.expect("every index was submitted").wait()).collect()
    }
}
