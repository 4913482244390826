/* ark355.h -- C ABI of libark355.so, the MI355X (gfx950) Groth16 prover backend.
 *
 * This is the drop-in boundary for the hot path of `ark_snark::SNARK::prove`
 * (/root/reference/snark/src/lib.rs:50-54): the Rust host keeps `ark-relations`' ConstraintSystem
 * (/root/reference/relations/src/gr1cs/constraint_system.rs) untouched, extracts
 *   - the R1CS matrices once per circuit   (to_matrices, constraint_system.rs:768-774;
 *                                           Matrix<F> = Vec<Vec<(F, usize)>>, utils/matrix.rs:4)
 *   - the full assignment z per proof      (instance_assignment || witness_assignment,
 *                                           constraint_system.rs:193-206, assignment.rs:11-21)
 * and calls the entry points below; see INTEGRATION.md for the Rust `extern "C"` block.
 *
 * Conventions:
 *  - every function returns int32: 0 = OK, negative = error (no exception crosses the boundary);
 *  - field elements are ark-ff memory images: little-endian u64 limbs, MONTGOMERY form
 *    (Fr: 32 B; Fq: 48 B BLS12-381 / 32 B BN254).  MSM scalars and r, s are CANONICAL
 *    (`into_bigint`) 32-byte little-endian integers;
 *  - G1 affine = x || y ; G2 affine = x.c0 || x.c1 || y.c0 || y.c1 ; the point at infinity is the
 *    all-zero encoding (the shim maps `Affine::infinity == true` to zeros and back);
 *  - the caller owns all host buffers; the library copies during the call and never retains host
 *    pointers.  Handles are opaque and freed by the matching *_free / *_destroy;
 *  - `*_dev` variants take DEVICE pointers (HIP allocations of the calling process).  The library reads them on
 *    its own streams: the CALLER must have completed (stream- or device-synchronised) whatever produced those
 *    buffers before the call.  Only ark355_ntt_fr_dev takes the producer's stream (void*, NULL = the context's own)
 *    and orders itself on it.
 */
#ifndef ARK355_H
#define ARK355_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ark355_ctx ark355_ctx;
typedef struct ark355_pk ark355_pk;
typedef struct ark355_r1cs ark355_r1cs;

enum { ARK355_BLS12_381 = 0, ARK355_BN254 = 1 };

/* error codes; the -16.. block mirrors ark_relations SynthesisError (utils/error.rs:5-21) */
enum {
  ARK355_OK = 0,
  ARK355_EINVAL = -1,
  ARK355_ENOMEM = -2,
  ARK355_EHIP = -3,
  ARK355_ERCCL = -4,
  ARK355_ENODEV = -5,
  ARK355_E_ASSIGNMENT_MISSING = -16,      /* SynthesisError::AssignmentMissing        (error.rs:9)  */
  ARK355_E_UNSATISFIABLE = -17,           /* SynthesisError::Unsatisfiable            (error.rs:14) */
  ARK355_E_POLY_DEGREE_TOO_LARGE = -18    /* SynthesisError::PolynomialDegreeTooLarge (error.rs:16) */
};

/* ---- context ------------------------------------------------------------------------------ */
int32_t ark355_ctx_create(int32_t device_id, ark355_ctx** out);
void ark355_ctx_destroy(ark355_ctx* ctx);
const char* ark355_last_error(const ark355_ctx* ctx);
/* library/ABI version: (major << 16) | minor */
uint32_t ark355_version(void);
/* sizes in bytes for `curve`: what[0]=Fr, [1]=Fq, [2]=G1 affine, [3]=G2 affine */
int32_t ark355_sizes(int32_t curve, uint32_t what[4]);

/* ---- runtime policy ------------------------------------------------------------------------------------------------
 * Every runtime switch of the library lives in ONE per-context struct (snark_amd/csrc/policy.h).  The environment
 * variables ARK355_<NAME> are read exactly once, by ark355_ctx_create; afterwards a context's policy changes only through
 * ark355_ctx_set_policy -- nothing on the proving path reads the environment, and two contexts of a process may differ
 * (A/B measurements inside one process).  Names (value = integer):
 *   per proof     SCHED (-1 measured choice [default], 0 one stream, 1 five-stream pipeline, 2 pipeline + epilogue stream
 *                 synchronises, 3 one stream + wait inside the HIP runtime), SCHED_EXPLORE (samples per schedule before the
 *                 measured choice latches; 0 = static defaults), WAIT_SPIN, WAIT_ADAPT, STREAM_PRIO, BATCH_TAILS,
 *                 SIDE_G2_TAILS, SIDE_WM, SIDE_G1_TAILS (a proof alone on the device: G2 tails / witness map / the tails of A, B1, L' on
 *                 side streams), SIDE_H_TAILS (pipeline: the last MSM's tails on the sort stream),
 *                 TRACE_HOST; legacy spellings SERIAL (1 -> SCHED 0, 0 -> SCHED 1) and EPILOGUE_SYNC (on a pipeline: 1 -> 2);
 *   per key load  MSM_C, MSM_C_H (window size of all tables / of the h_query table; 0 = planner), PACK_ROWS
 *                 (table rows bit-packed 1 / one word per limb 0 / per curve -1), TABLE_STRIDE, HBM_BUDGET_MB, SHARD_DIST_WM, RCCL_SELF (diagnostic: a rank at world size 1 exchanges with
 *                 itself) -- read when a key or base set is loaded THROUGH this context;
 *   per call      MSM_SEG, ACC_THREADS (workgroup size of the LDS-free accumulation kernels: 64 / 128 / 256; 0 [default]: 64 for a
 *                 proof alone on one stream, else 256), MSM_TWO_LEVEL_MIN, NTT_RMAX, NTT_DIRECT_MAX, NTT_NOFUSE (A/B and test knobs).
 *                 CHECK_SATISFIED (default 0: a proof is computed for whatever z is handed in, as ark-groth16's release build does;
 *                 1: ark355_prove / _dev / _batch also compare a_i b_i with c_i on the rows the witness map computes anyway -- the
 *                 check ark-groth16 runs under debug_assert!(cs.is_satisfied()) -- and return ARK355_E_UNSATISFIABLE, with the index
 *                 of the first unsatisfied constraint in ark355_last_error, instead of a proof that cannot verify).
 * ARK355_EINVAL for an unknown name.  ark355_prove_batch runs its worker contexts under the caller's policy.
 * (No counterpart in the reference: ark-groth16 has no runtime knobs; rayon's thread count is its only one.) */
int32_t ark355_ctx_set_policy(ark355_ctx* ctx, const char* name, int64_t value);
int32_t ark355_ctx_get_policy(ark355_ctx* ctx, const char* name, int64_t* value);

/* What the measured schedule choice (SCHED = -1) has seen for proofs shaped like `pk` on this context's device, for the
 * class "alone on the device" (in_flight = 0) or "other proofs in flight" (in_flight != 0): the schedule it latched
 * (-1 while still exploring / never run), per schedule the samples taken and their mean wall time in ms, and the
 * schedule the LAST proof of this context ran as.  ark355_sched_reset forgets the device's measurements. */
typedef struct {
  int32_t latched;
  int32_t last;
  uint32_t samples[4];
  double mean_ms[4];
} ark355_sched_report;
int32_t ark355_sched_info(const ark355_ctx* ctx, const ark355_pk* pk, int32_t in_flight, ark355_sched_report* out);
int32_t ark355_sched_reset(const ark355_ctx* ctx);
/* Diagnostic: which of the library's HIP streams share an in-order hardware queue of the runtime (streams on one queue
 * serialise whatever their events say).  Probes, on an IDLE device, the own streams of `count` (<= 16) contexts followed
 * by the three feeder streams of ctxs[0]'s pipeline: serialised is an (count + 3) x (count + 3) row-major matrix,
 * [i][j] = 1 when a kernel on stream j waited for a spinning kernel on stream i, 0 when it overtook it, -1 when the build
 * cannot measure it.  bench.py prints it; nothing on the proving path depends on it. */
int32_t ark355_diag_streams(ark355_ctx** ctxs, uint32_t count, int8_t* serialised);
/* Diagnostic: GPU-side cost of one dispatch in an in-order stream on this box: `launches` kernels that each spin for
 * `spin_us`, back to back; *gap_us = elapsed / launches - spin_us (a few us on most boxes, 50-90 us on some: see
 * DESIGN.md section 10).  *lanes (may be NULL) = the number of streams on pairwise different hardware queues the library
 * found for one-stream proofs on this device (probed once, at the first ark355_ctx_create). */
int32_t ark355_diag_dispatch(ark355_ctx* ctx, uint32_t launches, uint32_t spin_us, float* gap_us, uint32_t* lanes);
/* Diagnostic: the rate at which THIS box issues v_mad_u64_u32 right now -- the instruction the bucket-accumulation kernels are
 * made of, at their occupancy (two waves per SIMD), for about target_ms (0.1 .. 1000) milliseconds on the context's stream.
 * *tmad_per_s = 10^12 multiply-adds per second over the whole chip (negative: the build cannot measure it), *elapsed_ms (may be
 * NULL) = the duration of the measuring launch.  The sustained gfx clock under the power cap differs from box to box by a few
 * per cent; bench.py prices its integer roofline against this reading and reports box-normalised times with it. */
int32_t ark355_diag_mad_rate(ark355_ctx* ctx, float target_ms, float* tmad_per_s, float* elapsed_ms);
/* Diagnostic: the GPU's free-running counters, read by one wave per compute unit at (nearly) the same moment on the context's
 * stream (the call synchronises that stream): pairs[2 i] = shader-clock cycles (s_memtime: follows the gfx clock the power
 * management grants), pairs[2 i + 1] = ticks of the constant 100 MHz reference (s_memrealtime), for compute unit slot i < *count
 * (1024 slots; zeros where no wave landed; capacity = pairs the buffer holds, at least 1024).  The cycle counters of different
 * compute units differ by arbitrary offsets: two calls bracket a region, and per slot present in both
 * (cycles delta) / (ticks delta) x 100 MHz is the region's mean gfx clock; bench.py reports the median over the slots and
 * `gfx_cycles_per_constraint` with it. */
int32_t ark355_diag_clocks(ark355_ctx* ctx, uint64_t* pairs, uint32_t capacity, uint32_t* count);

/* Page-locked host memory for assignments / key vectors handed to the entry points below: H2D copies from pinned
 * memory run at PCIe rate (~55 GB/s) and truly asynchronously; pageable memory is staged by the runtime at a fraction
 * of that (a 32 MiB assignment at n = 2^20: ~0.6 ms pinned vs several ms pageable).  Optional: every entry point accepts
 * ordinary memory too. */
int32_t ark355_host_alloc(uint64_t bytes, void** out);
void ark355_host_free(void* p);

/* ---- proving key (replaces holding ark_groth16::ProvingKey<E> on the host; SNARK::ProvingKey,
 *      snark/src/lib.rs:25) ------------------------------------------------------------------- */
typedef struct {
  uint64_t num_instance;      /* ell, including the constant One (constraint_system.rs:218-220) */
  uint64_t num_witness;       /* w   (constraint_system.rs:223-225)                            */
  uint64_t domain_size;       /* N = next_pow2(n + ell); h_query holds N-1 points              */
  const uint8_t* a_query;     /* (ell+w) G1 */
  const uint8_t* b_g1_query;  /* (ell+w) G1 */
  const uint8_t* b_g2_query;  /* (ell+w) G2 */
  const uint8_t* h_query;     /* (N-1)   G1 */
  const uint8_t* l_query;     /* w       G1 */
  const uint8_t* alpha_g1;
  const uint8_t* beta_g1;
  const uint8_t* delta_g1;
  const uint8_t* beta_g2;
  const uint8_t* delta_g2;
} ark355_pk_desc;

int32_t ark355_pk_load(ark355_ctx* ctx, int32_t curve, const ark355_pk_desc* desc, ark355_pk** out);
void ark355_pk_free(ark355_pk* pk);

/* ---- R1CS matrices in CSR (replaces walking Matrix<F> rows on the host: mat_vec_mul,
 *      utils/matrix.rs:26-36; column convention Variable::get_variable_index,
 *      utils/variable.rs:105-113: 0 = One, 1..ell-1 = instance, ell.. = witness) --------------- */
int32_t ark355_r1cs_load(ark355_ctx* ctx, int32_t curve, uint64_t n_constraints, uint64_t num_instance,
                         uint64_t num_witness, const uint64_t* const row_ptr[3],
                         const uint32_t* const col[3], const uint8_t* const coeff[3],
                         ark355_r1cs** out);
void ark355_r1cs_free(ark355_r1cs* r1cs);
/* domain size N the library will use for this instance */
uint64_t ark355_r1cs_domain_size(const ark355_r1cs* r1cs);

/* ---- proof (SNARK::Proof, snark/src/lib.rs:32): affine A (G1), B (G2), C (G1), Montgomery raw.
 *      The arrays are sized for BLS12-381 (96 / 192 B); BN254 points (64 / 128 B) occupy the leading bytes of each
 *      array and the rest is zero -- ark355_sizes() gives the sizes in use. */
typedef struct {
  uint8_t a[96];
  uint8_t b[192];
  uint8_t c[96];
} ark355_proof_raw;

/* SNARK::prove hot path (snark/src/lib.rs:50-54; upstream create_proof_with_reduction_and_matrices):
 * z = full assignment (ell+w Fr, Montgomery), r/s canonical.  Returns ARK355_E_ASSIGNMENT_MISSING if
 * z_len < ell+w. */
int32_t ark355_prove(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1cs, const uint8_t* z,
                     uint64_t z_len, const uint8_t r[32], const uint8_t s[32], ark355_proof_raw* out);
/* same, z already resident in HBM */
int32_t ark355_prove_dev(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1cs, const void* d_z,
                         uint64_t z_len, const uint8_t r[32], const uint8_t s[32], ark355_proof_raw* out);

/* Many independent proofs of ONE circuit on one GPU (BASELINE.json configs[4]: 64 proofs at n = 2^18, eight per
 * GPU).  Replaces a pool of OS threads that each call SNARK::prove (snark/src/lib.rs:50-54) -- the reference's
 * ConstraintSystemRef is Rc<RefCell<..>> (constraint_system_ref.rs:33), so upstream parallelism is exactly "one
 * thread per proof" (Threading).  z[i]: host assignment of proof i (z_len Fr each, Montgomery);
 * r, s: count x 32 B canonical; out: count proofs.  Up to `inflight` (1..16) proofs are in flight on private
 * streams and scratch owned by `ctx`; the key and CSR handles are shared, read-only.  While one proof is in a
 * serial phase (digit sort, last bucket reduction, host finish) the others keep the CUs busy.  Returns the error
 * of the first failing proof (out[] of the others is still written). */
int32_t ark355_prove_batch(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1cs, const uint8_t* const* z,
                           uint64_t z_len, const uint8_t* r, const uint8_t* s, uint64_t count, uint32_t inflight,
                           ark355_proof_raw* out);

/* ---- one proof, MSM term ranges sharded over several GPUs (BASELINE.json configs[2]) ----------
 * Every rank loads shard `shard_index` of `shard_count` of the SAME key descriptor (terms
 * [T*i/G, T*(i+1)/G) of each query vector), computes the witness map redundantly, and returns the five
 * partial sums (A, B1, L', H as G1 XYZZ, then B2 as G2 XYZZ; ark355_partial_size() bytes).  The ranks
 * exchange the partials (an all-gather of <1 KiB per rank: RCCL has no elliptic-curve reduction operator) and
 * each calls ark355_prove_combine, which adds them and finishes the proof -- byte-identical to ark355_prove. */
int32_t ark355_pk_load_shard(ark355_ctx* ctx, int32_t curve, const ark355_pk_desc* desc, uint32_t shard_index,
                             uint32_t shard_count, ark355_pk** out);
uint64_t ark355_partial_size(int32_t curve);
int32_t ark355_prove_shard(ark355_ctx* ctx, const ark355_pk* pk_shard, const ark355_r1cs* r1cs, const uint8_t* z,
                           uint64_t z_len, const uint8_t r[32], const uint8_t s[32], uint8_t* out_partials);
int32_t ark355_prove_combine(ark355_ctx* ctx, int32_t curve, const uint8_t* partials, uint64_t count,
                             const uint8_t r[32], const uint8_t s[32], ark355_proof_raw* out);

/* ---- the exchange behind the ABI: RCCL communicator + collective sharded prove -----------------------------------
 * The reference's parallel unit is one OS thread per proof (ConstraintSystemRef is Rc<RefCell<..>>,
 * relations/src/gr1cs/constraint_system_ref.rs:33) behind SNARK::prove (snark/src/lib.rs:50-54); a single LARGE proof
 * (BASELINE.json configs[2]) is instead split here by MSM term ranges over the GPUs of a node, one host process per GPU.
 * Rank 0 obtains a communicator id (ark355_comm_unique_id) and hands it to the other ranks over whatever channel the
 * host already has (the Rust shim: its own IPC; the Python mirror: torch.distributed's store); every rank then calls
 * ark355_comm_init -- a collective, like ncclCommInitRank underneath.
 * ark355_prove_sharded is collective too: every rank passes its key shard (ark355_pk_load_shard with the rank as shard
 * index and the world size as shard count), the SAME full assignment z and the SAME r, s; every rank receives the same
 * proof, byte-identical to ark355_prove with the whole key.  `mode`:
 *   ARK355_SHARD_WINDOW       one ncclAllGather of the five XYZZ partial sums from HBM (default; latency-bound);
 *   ARK355_SHARD_BUCKET_RING  ring reduce-scatter of the bucket arrays (ncclSend/ncclRecv + EC-add kernel) before the
 *                             bucket reduction -- the literal "all-reduce of partial bucket sums" -- then the all-gather.
 * RCCL failures return ARK355_ERCCL (ark355_last_error has RCCL's message). */
typedef struct ark355_comm ark355_comm;
#define ARK355_COMM_ID_BYTES 128
enum { ARK355_SHARD_WINDOW = 0, ARK355_SHARD_BUCKET_RING = 1 };
int32_t ark355_comm_unique_id(uint8_t id[ARK355_COMM_ID_BYTES]);
int32_t ark355_comm_init(ark355_ctx* ctx, const uint8_t id[ARK355_COMM_ID_BYTES], int32_t rank, int32_t world,
                         ark355_comm** out);
void ark355_comm_destroy(ark355_comm* comm);
int32_t ark355_prove_sharded(ark355_ctx* ctx, ark355_comm* comm, const ark355_pk* pk_shard, const ark355_r1cs* r1cs,
                             const uint8_t* z, uint64_t z_len, const uint8_t r[32], const uint8_t s[32], int32_t mode,
                             ark355_proof_raw* out);
/* same, z already resident in this rank's HBM */
int32_t ark355_prove_sharded_dev(ark355_ctx* ctx, ark355_comm* comm, const ark355_pk* pk_shard, const ark355_r1cs* r1cs,
                                 const void* d_z, uint64_t z_len, const uint8_t r[32], const uint8_t s[32], int32_t mode,
                                 ark355_proof_raw* out);

/* ---- ark-serialize wire formats (SNARK::{ProvingKey, VerifyingKey, Proof}: CanonicalSerialize +
 *      CanonicalDeserialize, snark/src/lib.rs:25-36) ------------------------------------------------------------------
 * Point encodings as upstream writes them: BLS12-381 zcash/IETF (big-endian, flags in the first byte), BN254 ark-ec
 * SWFlags (little-endian, flags in the last byte); compressed != 0 selects the compressed form.  `validate`:
 *   ARK355_VALIDATE_FULL (1)   everything ark-serialize's Validate::Yes checks: reduced coordinates, the curve equation
 *                              (uncompressed form; compressed points are on the curve by construction) AND membership in
 *                              the prime-order subgroup ([r]P = O: BLS12-381 G1/G2, BN254 G2) -- use it for anything that
 *                              comes from an untrusted party (proofs: a small-order component vanishes in the pairing, so
 *                              without the test one proof has many accepted encodings);
 *   ARK355_VALIDATE_CURVE (2)  the same without the subgroup test (a 255-bit scalar multiplication per point): the
 *                              explicit opt-out for key material from a trusted source;
 *   ARK355_VALIDATE_NONE (0)   Validate::No.
 * Flag combinations upstream rejects in every mode are rejected in every mode here: BLS12-381 sort bit without the
 * compressed bit or together with the infinity bit, a compressed bit that does not match the requested form
 * (ark-bls12-381 EncodingFlags::get_flags); BN254 both flag bits set (ark-ec SWFlags::from_u8).  Bytes under an infinity
 * flag, in every mode as upstream: BLS12-381 must be all zero (its point readers refuse anything else), BN254 must be
 * reduced field elements and are otherwise ignored (ark-ec's generic reader returns the identity). */
#define ARK355_VALIDATE_NONE 0
#define ARK355_VALIDATE_FULL 1
#define ARK355_VALIDATE_CURVE 2
/* bytes of one encoded point of `group` (1 | 2) */
uint64_t ark355_point_size(int32_t curve, int32_t group, int32_t compressed);
/* the byte stream of an ark_groth16::ProvingKey<E> (vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query,
 * l_query) -> resident key, exactly as ark355_pk_load would build it from the decoded vectors.  The points are decoded
 * on the device (one lane per point; the compressed form costs a square root each).  ARK355_EINVAL for truncated /
 * inconsistent streams and bad points (ark355_last_error names the vector and index). */
int32_t ark355_pk_load_bytes(ark355_ctx* ctx, int32_t curve, const uint8_t* bytes, uint64_t len, int32_t compressed,
                             int32_t validate, ark355_pk** out);
/* dimensions of a resident key: num_instance (ell), num_witness (w), domain size N */
int32_t ark355_pk_dims(const ark355_pk* pk, uint64_t* num_instance, uint64_t* num_witness, uint64_t* domain_size);
/* How a resident key sits in HBM: Pippenger window size c, windows per scalar, window stride of its tables
 * (1 = a table per window: one bucket set, no doublings; s > 1 = every s-th window has a table and an MSM keeps s
 * bucket sets -- chosen at load time as the smallest stride whose tables fit the device next to the provers' scratch;
 * ARK355_ENOMEM from the load when none does) and the bytes its five window tables occupy.  Any pointer may be NULL.
 * (No counterpart in the reference: ark-groth16 keeps `ProvingKey<E>` as plain vectors in host memory.) */
int32_t ark355_pk_table_info(const ark355_pk* pk, uint32_t* window_bits, uint32_t* windows, uint32_t* table_stride,
                             uint64_t* table_bytes);
/* n encoded points <-> n raw affine images (x || y Montgomery; the layout of every other entry point), on the device */
int32_t ark355_points_decode(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* in, uint64_t n,
                             int32_t compressed, int32_t validate, uint8_t* out_raw);
int32_t ark355_points_encode(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* in_raw, uint64_t n,
                             int32_t compressed, uint8_t* out);
/* Proof = a || b || c; out must hold 2 * point_size(G1) + point_size(G2) bytes */
int32_t ark355_proof_to_bytes(int32_t curve, const ark355_proof_raw* proof, int32_t compressed, uint8_t* out);
int32_t ark355_proof_from_bytes(int32_t curve, const uint8_t* in, uint64_t len, int32_t compressed, int32_t validate,
                                ark355_proof_raw* out);

/* ---- building blocks ---------------------------------------------------------------------- */
/* R1CS -> QAP witness map h[0..N) (Montgomery), SURVEY Appendix A steps 1-5 */
int32_t ark355_witness_map(ark355_ctx* ctx, const ark355_r1cs* r1cs, const uint8_t* z, uint64_t z_len,
                           uint8_t* h_out);
/* The same map computed the way the `world` GPUs of a sharded proof compute it (every rank owns 1/world of each vector,
 * three all-to-all exchanges; snark_amd/csrc/witness_dist_impl.cuh), with ALL ranks on this one device and
 * device-to-device copies as the exchange: the test / diagnostic entry of the distributed path on a single GPU
 * (ark355_prove_sharded runs it over RCCL).  world: a power of two >= 2 with 8 * world^2 <= N; ARK355_EINVAL otherwise. */
int32_t ark355_witness_map_dist_sim(ark355_ctx* ctx, const ark355_r1cs* r1cs, const uint8_t* z, uint64_t z_len,
                                    uint32_t world, uint8_t* h_out);
/* which_is_unsatisfied (constraint_system.rs:661-687) for the R1CS predicate: first_bad = -1 if
 * satisfied, else the first constraint index with <A_i,z>*<B_i,z> != <C_i,z> */
int32_t ark355_is_satisfied(ark355_ctx* ctx, const ark355_r1cs* r1cs, const uint8_t* z, uint64_t z_len,
                            int64_t* first_bad);
/* A z, B z, C z (n Fr each, Montgomery): mat_vec_mul, utils/matrix.rs:26-36 */
int32_t ark355_r1cs_mat_vec(ark355_ctx* ctx, const ark355_r1cs* r1cs, const uint8_t* z, uint64_t z_len,
                            uint8_t* az, uint8_t* bz, uint8_t* cz);

/* in-place radix-2 NTT over Fr, natural order in and out (ark-poly Radix2EvaluationDomain
 * fft / ifft / coset_fft / coset_ifft).  Errors with ARK355_E_POLY_DEGREE_TOO_LARGE past the
 * field's two-adicity. */
int32_t ark355_ntt_fr(ark355_ctx* ctx, int32_t curve, uint8_t* data, uint32_t log_n, int32_t inverse,
                      int32_t coset);
int32_t ark355_ntt_fr_dev(ark355_ctx* ctx, int32_t curve, void* d_data, void* d_scratch, uint32_t log_n,
                          int32_t inverse, int32_t coset, void* stream);

/* sum_i scalars[i] * bases[i] -> affine (ark-ec VariableBaseMSM::msm_bigint semantics: canonical
 * scalars; zero scalars and infinity bases contribute nothing) */
int32_t ark355_msm_g1(ark355_ctx* ctx, int32_t curve, const uint8_t* bases, const uint8_t* scalars,
                      uint64_t n, uint8_t* out_affine);
int32_t ark355_msm_g2(ark355_ctx* ctx, int32_t curve, const uint8_t* bases, const uint8_t* scalars,
                      uint64_t n, uint8_t* out_affine);

/* device-resident MSM: bases are uploaded once into a handle, scalars live in HBM */
typedef struct ark355_bases ark355_bases;
int32_t ark355_bases_load(ark355_ctx* ctx, int32_t curve, int32_t group /*1|2*/, const uint8_t* bases,
                          uint64_t n, ark355_bases** out);
void ark355_bases_free(ark355_bases* b);
/* scalars_mont != 0: the scalars are Montgomery Fr images and are converted on device */
int32_t ark355_msm_dev(ark355_ctx* ctx, const ark355_bases* bases, const void* d_scalars, uint64_t n,
                       int32_t scalars_mont, uint8_t* out_affine);
/* partial result as raw XYZZ (4 coordinates, Montgomery) for cross-GPU combination */
int32_t ark355_msm_dev_partial(ark355_ctx* ctx, const ark355_bases* bases, const void* d_scalars,
                               uint64_t n, int32_t scalars_mont, uint8_t* out_xyzz);
/* sum of `count` raw XYZZ partials -> affine (the local EC-add after the RCCL all-gather) */
int32_t ark355_xyzz_sum(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* partials,
                        uint64_t count, uint8_t* out_affine);

/* out[i] = scalars[i] * base (fixed base, canonical scalars) -> affine; used by
 * circuit_specific_setup (snark/src/lib.rs:43-46) to build the query vectors */
int32_t ark355_fixed_base_mul(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* base,
                              const uint8_t* scalars, uint64_t n, uint8_t* out_affine);

/* ---- batch verification (SNARK::verify / verify_with_processed_vk, snark/src/lib.rs:59-80) ------
 * Checks `count` proofs of ONE verifying key at once with the random-linear-combination test
 *   prod_j e(rho_j A_j, B_j) = e((sum rho_j) alpha, beta) e(sum_i (sum_j rho_j x_ji) gamma_abc_i, gamma) e(sum rho_j C_j, delta):
 * count + 3 Miller loops and ONE final exponentiation instead of 4 count pairings.  The two multi-scalar sums run on
 * the device; the Miller loops and the final exponentiation on host threads.  public_inputs: count x (num_instance - 1)
 * Fr (Montgomery; the leading One is implicit, as in SNARK::verify); rho: count x 32 B canonical, non-zero, drawn by
 * the caller from its rng (soundness error ~ 1/|rho|); NULL is allowed for count == 1 (plain verification).
 * *ok = 1 iff every proof verifies (with overwhelming probability over rho). */
typedef struct {
  uint64_t num_instance;        /* ell = gamma_abc_g1 length */
  const uint8_t* alpha_g1;
  const uint8_t* beta_g2;
  const uint8_t* gamma_g2;
  const uint8_t* delta_g2;
  const uint8_t* gamma_abc_g1;  /* ell G1 */
} ark355_vk_desc;
int32_t ark355_verify_batch(ark355_ctx* ctx, int32_t curve, const ark355_vk_desc* vk, const ark355_proof_raw* proofs,
                            const uint8_t* public_inputs, const uint8_t* rho, uint64_t count, int32_t* ok);

/* The scalars of the Groth16 generator (circuit_specific_setup, snark/src/lib.rs:43-46; upstream
 * generate_parameters_with_qap) from the R1CS matrices in CSR and the five trapdoor elements tau, alpha, beta, gamma,
 * delta (5 x 32 B canonical): u_j(tau), v_j(tau), w_j(tau) (num_instance + num_witness each), l_j (num_witness),
 * gamma_abc_j (num_instance), h_i (N - 1) -- canonical 32-byte values ready for ark355_fixed_base_mul.  Host threads
 * only (the library's own field code); ARK355_E_POLY_DEGREE_TOO_LARGE past the field's two-adicity. */
int32_t ark355_setup_scalars(int32_t curve, uint64_t n_constraints, uint64_t num_instance, uint64_t num_witness,
                             const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                             const uint8_t* const coeff[3], const uint8_t* trapdoor, uint8_t* out_u, uint8_t* out_v,
                             uint8_t* out_w, uint8_t* out_l, uint8_t* out_gamma_abc, uint8_t* out_h);

/* ---- timings of the last prove on this context (ms, measured with HIP events) --------------- */
typedef struct {
  float total_ms;
  float h2d_ms;
  float witness_map_ms;
  float msm_h_ms;
  float msm_l_ms;
  float msm_ab_g1_ms;
  float msm_b_g2_ms;
  float finalize_ms;
} ark355_timings;
int32_t ark355_get_timings(const ark355_ctx* ctx, ark355_timings* out);

/* per-kernel HIP-event timing of the dominant kernel (bucket accumulation) of the last MSM/prove
 * on this context: sum of launch durations (ms), number of launches, points accumulated */
int32_t ark355_get_kernel_stats(const ark355_ctx* ctx, float* accumulate_ms, uint64_t* launches,
                                uint64_t* points);

#ifdef __cplusplus
}
#endif
#endif /* ARK355_H */
