//! `extern "C"` image of `include/ark355.h` -- one declaration per entry point, same order, same argument lists.
//! tests/test_rust_shim.py parses both files and fails when they drift apart.
#![allow(non_camel_case_types, dead_code)]

use core::ffi::{c_char, c_void};

#[repr(C)]
pub struct ark355_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct ark355_pk {
    _p: [u8; 0],
}
#[repr(C)]
pub struct ark355_r1cs {
    _p: [u8; 0],
}
#[repr(C)]
pub struct ark355_comm {
    _p: [u8; 0],
}
#[repr(C)]
pub struct ark355_bases {
    _p: [u8; 0],
}

pub const ARK355_BLS12_381: i32 = 0;
pub const ARK355_BN254: i32 = 1;

pub const ARK355_OK: i32 = 0;
pub const ARK355_EINVAL: i32 = -1;
pub const ARK355_ENOMEM: i32 = -2;
pub const ARK355_EHIP: i32 = -3;
pub const ARK355_ERCCL: i32 = -4;
pub const ARK355_ENODEV: i32 = -5;
pub const ARK355_E_ASSIGNMENT_MISSING: i32 = -16;
pub const ARK355_E_UNSATISFIABLE: i32 = -17;
pub const ARK355_E_POLY_DEGREE_TOO_LARGE: i32 = -18;

pub const ARK355_COMM_ID_BYTES: usize = 128;
pub const ARK355_SHARD_WINDOW: i32 = 0;
pub const ARK355_SHARD_BUCKET_RING: i32 = 1;
/// `validate` of the wire-format entry points: Validate::No / Validate::Yes (curve + prime-order subgroup) / curve only
pub const ARK355_VALIDATE_NONE: i32 = 0;
pub const ARK355_VALIDATE_FULL: i32 = 1;
pub const ARK355_VALIDATE_CURVE: i32 = 2;

#[repr(C)]
pub struct ark355_pk_desc {
    pub num_instance: u64,
    pub num_witness: u64,
    pub domain_size: u64,
    pub a_query: *const u8,
    pub b_g1_query: *const u8,
    pub b_g2_query: *const u8,
    pub h_query: *const u8,
    pub l_query: *const u8,
    pub alpha_g1: *const u8,
    pub beta_g1: *const u8,
    pub delta_g1: *const u8,
    pub beta_g2: *const u8,
    pub delta_g2: *const u8,
}

#[repr(C)]
pub struct ark355_vk_desc {
    pub num_instance: u64,
    pub alpha_g1: *const u8,
    pub beta_g2: *const u8,
    pub gamma_g2: *const u8,
    pub delta_g2: *const u8,
    pub gamma_abc_g1: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ark355_proof_raw {
    pub a: [u8; 96],
    pub b: [u8; 192],
    pub c: [u8; 96],
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct ark355_timings {
    pub total_ms: f32,
    pub h2d_ms: f32,
    pub witness_map_ms: f32,
    pub msm_h_ms: f32,
    pub msm_l_ms: f32,
    pub msm_ab_g1_ms: f32,
    pub msm_b_g2_ms: f32,
    pub finalize_ms: f32,
}

/// What the measured schedule choice has seen for one (proof shape, alone | in flight) class: `ark355_sched_info`.
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct ark355_sched_report {
    pub latched: i32,
    pub last: i32,
    pub samples: [u32; 4],
    pub mean_ms: [f64; 4],
}

extern "C" {
    pub fn ark355_ctx_create(device_id: i32, out: *mut *mut ark355_ctx) -> i32;
    pub fn ark355_ctx_destroy(ctx: *mut ark355_ctx);
    pub fn ark355_last_error(ctx: *const ark355_ctx) -> *const c_char;
    pub fn ark355_version() -> u32;
    pub fn ark355_sizes(curve: i32, what: *mut u32) -> i32;

    pub fn ark355_ctx_set_policy(ctx: *mut ark355_ctx, name: *const c_char, value: i64) -> i32;
    pub fn ark355_ctx_get_policy(ctx: *mut ark355_ctx, name: *const c_char, value: *mut i64) -> i32;
    pub fn ark355_sched_info(ctx: *const ark355_ctx, pk: *const ark355_pk, in_flight: i32, out: *mut ark355_sched_report) -> i32;
    pub fn ark355_sched_reset(ctx: *const ark355_ctx) -> i32;
    pub fn ark355_diag_streams(ctxs: *mut *mut ark355_ctx, count: u32, serialised: *mut i8) -> i32;
    pub fn ark355_diag_dispatch(ctx: *mut ark355_ctx, launches: u32, spin_us: u32, gap_us: *mut f32, lanes: *mut u32) -> i32;
    pub fn ark355_diag_mad_rate(ctx: *mut ark355_ctx, target_ms: f32, tmad_per_s: *mut f32, elapsed_ms: *mut f32) -> i32;
    pub fn ark355_diag_clocks(ctx: *mut ark355_ctx, pairs: *mut u64, capacity: u32, count: *mut u32) -> i32;

    pub fn ark355_host_alloc(bytes: u64, out: *mut *mut c_void) -> i32;
    pub fn ark355_host_free(p: *mut c_void);

    pub fn ark355_pk_load(ctx: *mut ark355_ctx, curve: i32, desc: *const ark355_pk_desc, out: *mut *mut ark355_pk) -> i32;
    pub fn ark355_pk_free(pk: *mut ark355_pk);

    pub fn ark355_r1cs_load(
        ctx: *mut ark355_ctx,
        curve: i32,
        n_constraints: u64,
        num_instance: u64,
        num_witness: u64,
        row_ptr: *const *const u64,
        col: *const *const u32,
        coeff: *const *const u8,
        out: *mut *mut ark355_r1cs,
    ) -> i32;
    pub fn ark355_r1cs_free(r1cs: *mut ark355_r1cs);
    pub fn ark355_r1cs_domain_size(r1cs: *const ark355_r1cs) -> u64;

    pub fn ark355_prove(
        ctx: *mut ark355_ctx,
        pk: *const ark355_pk,
        r1cs: *const ark355_r1cs,
        z: *const u8,
        z_len: u64,
        r: *const u8,
        s: *const u8,
        out: *mut ark355_proof_raw,
    ) -> i32;
    pub fn ark355_prove_dev(
        ctx: *mut ark355_ctx,
        pk: *const ark355_pk,
        r1cs: *const ark355_r1cs,
        d_z: *const c_void,
        z_len: u64,
        r: *const u8,
        s: *const u8,
        out: *mut ark355_proof_raw,
    ) -> i32;
    pub fn ark355_prove_batch(
        ctx: *mut ark355_ctx,
        pk: *const ark355_pk,
        r1cs: *const ark355_r1cs,
        z: *const *const u8,
        z_len: u64,
        r: *const u8,
        s: *const u8,
        count: u64,
        inflight: u32,
        out: *mut ark355_proof_raw,
    ) -> i32;

    pub fn ark355_pk_load_shard(
        ctx: *mut ark355_ctx,
        curve: i32,
        desc: *const ark355_pk_desc,
        shard_index: u32,
        shard_count: u32,
        out: *mut *mut ark355_pk,
    ) -> i32;
    pub fn ark355_partial_size(curve: i32) -> u64;
    pub fn ark355_prove_shard(
        ctx: *mut ark355_ctx,
        pk_shard: *const ark355_pk,
        r1cs: *const ark355_r1cs,
        z: *const u8,
        z_len: u64,
        r: *const u8,
        s: *const u8,
        out_partials: *mut u8,
    ) -> i32;
    pub fn ark355_prove_combine(
        ctx: *mut ark355_ctx,
        curve: i32,
        partials: *const u8,
        count: u64,
        r: *const u8,
        s: *const u8,
        out: *mut ark355_proof_raw,
    ) -> i32;

    pub fn ark355_comm_unique_id(id: *mut u8) -> i32;
    pub fn ark355_comm_init(ctx: *mut ark355_ctx, id: *const u8, rank: i32, world: i32, out: *mut *mut ark355_comm) -> i32;
    pub fn ark355_comm_destroy(comm: *mut ark355_comm);
    pub fn ark355_prove_sharded(
        ctx: *mut ark355_ctx,
        comm: *mut ark355_comm,
        pk_shard: *const ark355_pk,
        r1cs: *const ark355_r1cs,
        z: *const u8,
        z_len: u64,
        r: *const u8,
        s: *const u8,
        mode: i32,
        out: *mut ark355_proof_raw,
    ) -> i32;
    pub fn ark355_prove_sharded_dev(
        ctx: *mut ark355_ctx,
        comm: *mut ark355_comm,
        pk_shard: *const ark355_pk,
        r1cs: *const ark355_r1cs,
        d_z: *const c_void,
        z_len: u64,
        r: *const u8,
        s: *const u8,
        mode: i32,
        out: *mut ark355_proof_raw,
    ) -> i32;

    pub fn ark355_point_size(curve: i32, group: i32, compressed: i32) -> u64;
    pub fn ark355_pk_load_bytes(
        ctx: *mut ark355_ctx,
        curve: i32,
        bytes: *const u8,
        len: u64,
        compressed: i32,
        validate: i32,
        out: *mut *mut ark355_pk,
    ) -> i32;
    pub fn ark355_pk_dims(pk: *const ark355_pk, num_instance: *mut u64, num_witness: *mut u64, domain_size: *mut u64) -> i32;
    pub fn ark355_pk_table_info(pk: *const ark355_pk, window_bits: *mut u32, windows: *mut u32, table_stride: *mut u32, table_bytes: *mut u64) -> i32;
    pub fn ark355_points_decode(
        ctx: *mut ark355_ctx,
        curve: i32,
        group: i32,
        input: *const u8,
        n: u64,
        compressed: i32,
        validate: i32,
        out_raw: *mut u8,
    ) -> i32;
    pub fn ark355_points_encode(
        ctx: *mut ark355_ctx,
        curve: i32,
        group: i32,
        in_raw: *const u8,
        n: u64,
        compressed: i32,
        out: *mut u8,
    ) -> i32;
    pub fn ark355_proof_to_bytes(curve: i32, proof: *const ark355_proof_raw, compressed: i32, out: *mut u8) -> i32;
    pub fn ark355_proof_from_bytes(
        curve: i32,
        input: *const u8,
        len: u64,
        compressed: i32,
        validate: i32,
        out: *mut ark355_proof_raw,
    ) -> i32;

    pub fn ark355_witness_map(ctx: *mut ark355_ctx, r1cs: *const ark355_r1cs, z: *const u8, z_len: u64, h_out: *mut u8) -> i32;
    pub fn ark355_witness_map_dist_sim(ctx: *mut ark355_ctx, r1cs: *const ark355_r1cs, z: *const u8, z_len: u64, world: u32, h_out: *mut u8) -> i32;
    pub fn ark355_is_satisfied(ctx: *mut ark355_ctx, r1cs: *const ark355_r1cs, z: *const u8, z_len: u64, first_bad: *mut i64) -> i32;
    pub fn ark355_r1cs_mat_vec(
        ctx: *mut ark355_ctx,
        r1cs: *const ark355_r1cs,
        z: *const u8,
        z_len: u64,
        az: *mut u8,
        bz: *mut u8,
        cz: *mut u8,
    ) -> i32;

    pub fn ark355_ntt_fr(ctx: *mut ark355_ctx, curve: i32, data: *mut u8, log_n: u32, inverse: i32, coset: i32) -> i32;
    pub fn ark355_ntt_fr_dev(
        ctx: *mut ark355_ctx,
        curve: i32,
        d_data: *mut c_void,
        d_scratch: *mut c_void,
        log_n: u32,
        inverse: i32,
        coset: i32,
        stream: *mut c_void,
    ) -> i32;

    pub fn ark355_msm_g1(ctx: *mut ark355_ctx, curve: i32, bases: *const u8, scalars: *const u8, n: u64, out_affine: *mut u8) -> i32;
    pub fn ark355_msm_g2(ctx: *mut ark355_ctx, curve: i32, bases: *const u8, scalars: *const u8, n: u64, out_affine: *mut u8) -> i32;

    pub fn ark355_bases_load(ctx: *mut ark355_ctx, curve: i32, group: i32, bases: *const u8, n: u64, out: *mut *mut ark355_bases) -> i32;
    pub fn ark355_bases_free(b: *mut ark355_bases);
    pub fn ark355_msm_dev(
        ctx: *mut ark355_ctx,
        bases: *const ark355_bases,
        d_scalars: *const c_void,
        n: u64,
        scalars_mont: i32,
        out_affine: *mut u8,
    ) -> i32;
    pub fn ark355_msm_dev_partial(
        ctx: *mut ark355_ctx,
        bases: *const ark355_bases,
        d_scalars: *const c_void,
        n: u64,
        scalars_mont: i32,
        out_xyzz: *mut u8,
    ) -> i32;
    pub fn ark355_xyzz_sum(ctx: *mut ark355_ctx, curve: i32, group: i32, partials: *const u8, count: u64, out_affine: *mut u8) -> i32;

    pub fn ark355_fixed_base_mul(
        ctx: *mut ark355_ctx,
        curve: i32,
        group: i32,
        base: *const u8,
        scalars: *const u8,
        n: u64,
        out_affine: *mut u8,
    ) -> i32;

    pub fn ark355_verify_batch(
        ctx: *mut ark355_ctx,
        curve: i32,
        vk: *const ark355_vk_desc,
        proofs: *const ark355_proof_raw,
        public_inputs: *const u8,
        rho: *const u8,
        count: u64,
        ok: *mut i32,
    ) -> i32;

    pub fn ark355_setup_scalars(
        curve: i32,
        n_constraints: u64,
        num_instance: u64,
        num_witness: u64,
        row_ptr: *const *const u64,
        col: *const *const u32,
        coeff: *const *const u8,
        trapdoor: *const u8,
        out_u: *mut u8,
        out_v: *mut u8,
        out_w: *mut u8,
        out_l: *mut u8,
        out_gamma_abc: *mut u8,
        out_h: *mut u8,
    ) -> i32;

    pub fn ark355_get_timings(ctx: *const ark355_ctx, out: *mut ark355_timings) -> i32;
    pub fn ark355_get_kernel_stats(ctx: *const ark355_ctx, accumulate_ms: *mut f32, launches: *mut u64, points: *mut u64) -> i32;
}
