//! hal_hip.rs — `impl Hal for HipHal`, `impl Buffer<T> for HipBuffer<T>` and `impl CircuitHal<HipHal> for HipCircuitHal` over
//! risc0-sys-hip (libzkhal_mi355x.so): the file a maintainer adds as `risc0-zkp/src/hal/hip.rs` behind `feature = "hip"`, next to
//! `hal/cuda.rs`.  It is what `default_prover().prove(env, elf)` (/root/reference/crates/host/src/lib.rs:137) ends up driving,
//! through `risc0_zkp::prove::Prover<'a, H: Hal>` inside the r0vm server.
//!
//! Written against the trait as RECALLED from risc0-zkp 3.0.2 `src/hal/mod.rs` (un-vendored: /root/reference/Cargo.lock:5393).
//! No Rust toolchain exists in the build image: this file has NOT been compiled.  What IS checked mechanically
//! (tests/test_rust_shim.py): every `zkh_*` call below names a function of the generated extern block with the right number of
//! arguments, and every method of the recalled `Hal` / `Buffer` / `CircuitHal` traits has a body here — none is elided.
//!
//! Conventions: `Elem` = 1 word, `ExtElem` = 4 words, `Digest` = 8 words, all raw Montgomery `u32`s exactly as upstream stores
//! them (`bytemuck::cast_slice` is enough in both directions).  One `HipHal` = one GPU + one in-order HIP stream, driven by the
//! single prover thread (upstream HALs are `!Sync` in practice too); `Buffer` handles are reference counted inside the library.

use std::cell::RefCell;
use std::ffi::CString;
use std::fmt::Debug;
use std::marker::PhantomData;
use std::rc::Rc;

use bytemuck::Pod;
use risc0_core::field::baby_bear::{BabyBear, BabyBearElem, BabyBearExtElem};
use risc0_sys_hip as sys;
use risc0_sys_hip::{ffi_wrap, ZkhBuf, ZkhCircuit, ZkhCtx};

use crate::core::digest::Digest;
use crate::core::hash::{poseidon2::Poseidon2HashSuite, HashSuite};
use crate::hal::{Buffer, CircuitHal, Hal};
use crate::INV_RATE;

/// Words per element of the three payload types (`u32` itself: 1).
fn words_of<T>() -> usize {
    std::mem::size_of::<T>() / 4
}

fn ffi<F: FnOnce() -> *const std::os::raw::c_char>(f: F) {
    // upstream's CUDA HAL panics on a failed kernel too (`ffi_wrap(..).unwrap()`): a HAL op has no error channel in the trait
    ffi_wrap(f).unwrap()
}

struct CtxHandle(*mut ZkhCtx);
impl Drop for CtxHandle {
    fn drop(&mut self) {
        unsafe { sys::zkh_ctx_destroy(self.0) }
    }
}

/// `impl Hal`: one MI355X.
pub struct HipHal {
    ctx: Rc<CtxHandle>,
    suite: HashSuite<BabyBear>,
}

/// `impl Buffer<T>`: a reference-counted view {allocation, offset, length} inside the library.
pub struct HipBuffer<T> {
    raw: *mut ZkhBuf,
    ctx: Rc<CtxHandle>,
    name: &'static str,
    _t: PhantomData<T>,
}

impl<T> Clone for HipBuffer<T> {
    fn clone(&self) -> Self {
        unsafe { sys::zkh_retain(self.raw) };
        HipBuffer { raw: self.raw, ctx: self.ctx.clone(), name: self.name, _t: PhantomData }
    }
}

impl<T> Drop for HipBuffer<T> {
    fn drop(&mut self) {
        unsafe { sys::zkh_release(self.raw) }
    }
}

impl<T: Pod> HipBuffer<T> {
    fn read_words(&self, off_items: usize, n_items: usize) -> Vec<u32> {
        let w = words_of::<T>();
        let mut host = vec![0u32; n_items * w];
        ffi(|| unsafe { sys::zkh_read(self.ctx.0, self.raw, host.as_mut_ptr(), off_items * w, n_items * w) });
        host
    }
}

impl<T: Pod + Clone> Buffer<T> for HipBuffer<T> {
    fn name(&self) -> &'static str {
        self.name
    }

    fn size(&self) -> usize {
        unsafe { sys::zkh_size(self.raw) } / words_of::<T>()
    }

    fn slice(&self, offset: usize, size: usize) -> Self {
        let w = words_of::<T>();
        let mut out: *mut ZkhBuf = std::ptr::null_mut();
        ffi(|| unsafe { sys::zkh_slice(self.raw, offset * w, size * w, &mut out) });
        HipBuffer { raw: out, ctx: self.ctx.clone(), name: self.name, _t: PhantomData }
    }

    fn get_at(&self, idx: usize) -> T {
        let words = self.read_words(idx, 1);
        bytemuck::cast_slice::<u32, T>(&words)[0]
    }

    fn view<F: FnOnce(&[T])>(&self, f: F) {
        let words = self.read_words(0, self.size()); // synchronises the stream, then D2H
        f(bytemuck::cast_slice::<u32, T>(&words))
    }

    fn view_mut<F: FnOnce(&mut [T])>(&self, f: F) {
        let mut words = self.read_words(0, self.size());
        f(bytemuck::cast_slice_mut::<u32, T>(&mut words));
        ffi(|| unsafe { sys::zkh_write(self.ctx.0, self.raw, words.as_ptr(), 0, words.len()) });
    }

    fn to_vec(&self) -> Vec<T> {
        let words = self.read_words(0, self.size());
        bytemuck::cast_slice::<u32, T>(&words).to_vec()
    }
}

impl HipHal {
    /// `CudaHal::new` analogue: binds `device_ordinal` (one r0vm worker per GPU: HIP_VISIBLE_DEVICES=k, ordinal 0).
    pub fn new(device_ordinal: i32) -> Self {
        let suite_name = CString::new("poseidon2").unwrap();
        let mut ctx: *mut ZkhCtx = std::ptr::null_mut();
        ffi(|| unsafe { sys::zkh_ctx_create(device_ordinal, suite_name.as_ptr(), &mut ctx) });
        // bind the prover thread (and the pinned blocks it allocates) next to the GPU's root port; a no-op where the host reports no node
        ffi(|| unsafe { sys::zkh_bind_thread_to_device(device_ordinal, 0, 1, std::ptr::null_mut(), std::ptr::null_mut()) });
        HipHal { ctx: Rc::new(CtxHandle(ctx)), suite: Poseidon2HashSuite::new_suite() }
    }

    fn alloc<T: Pod>(&self, name: &'static str, items: usize, zero: bool) -> HipBuffer<T> {
        let cname = CString::new(name).unwrap();
        let mut out: *mut ZkhBuf = std::ptr::null_mut();
        ffi(|| unsafe { sys::zkh_alloc(self.ctx.0, cname.as_ptr(), items * words_of::<T>(), zero as i32, &mut out) });
        HipBuffer { raw: out, ctx: self.ctx.clone(), name, _t: PhantomData }
    }

    fn copy_from<T: Pod>(&self, name: &'static str, slice: &[T]) -> HipBuffer<T> {
        let cname = CString::new(name).unwrap();
        let words: &[u32] = bytemuck::cast_slice(slice);
        let mut out: *mut ZkhBuf = std::ptr::null_mut();
        ffi(|| unsafe { sys::zkh_copy_from(self.ctx.0, cname.as_ptr(), words.as_ptr(), words.len(), &mut out) });
        HipBuffer { raw: out, ctx: self.ctx.clone(), name, _t: PhantomData }
    }

    fn ext_words(e: &BabyBearExtElem) -> *const u32 {
        e as *const BabyBearExtElem as *const u32 // 4 Montgomery words, the layout `ExtElem::to_u32_words` has
    }
}

impl Hal for HipHal {
    type Field = BabyBear;
    type Elem = BabyBearElem;
    type ExtElem = BabyBearExtElem;
    type Buffer<T: Clone + Debug + PartialEq> = HipBuffer<T>;

    fn has_unified_memory(&self) -> bool {
        false
    }

    fn get_hash_suite(&self) -> &HashSuite<Self::Field> {
        &self.suite // the Fiat-Shamir sponge / WriteIOP stay upstream's host code
    }

    fn alloc_digest(&self, name: &'static str, size: usize) -> Self::Buffer<Digest> {
        self.alloc(name, size, false)
    }

    fn alloc_elem(&self, name: &'static str, size: usize) -> Self::Buffer<Self::Elem> {
        self.alloc(name, size, false)
    }

    fn alloc_elem_init(&self, name: &'static str, size: usize, value: Self::Elem) -> Self::Buffer<Self::Elem> {
        self.copy_from(name, &vec![value; size])
    }

    fn alloc_extelem(&self, name: &'static str, size: usize) -> Self::Buffer<Self::ExtElem> {
        self.alloc(name, size, false)
    }

    fn alloc_extelem_zeroed(&self, name: &'static str, size: usize) -> Self::Buffer<Self::ExtElem> {
        self.alloc(name, size, true) // Montgomery zero is the all-zero word
    }

    fn alloc_u32(&self, name: &'static str, size: usize) -> Self::Buffer<u32> {
        self.alloc(name, size, false)
    }

    fn copy_from_digest(&self, name: &'static str, slice: &[Digest]) -> Self::Buffer<Digest> {
        self.copy_from(name, slice)
    }

    fn copy_from_elem(&self, name: &'static str, slice: &[Self::Elem]) -> Self::Buffer<Self::Elem> {
        self.copy_from(name, slice)
    }

    fn copy_from_extelem(&self, name: &'static str, slice: &[Self::ExtElem]) -> Self::Buffer<Self::ExtElem> {
        self.copy_from(name, slice)
    }

    fn copy_from_u32(&self, name: &'static str, slice: &[u32]) -> Self::Buffer<u32> {
        self.copy_from(name, slice)
    }

    fn batch_expand_into_evaluate_ntt(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::Elem>, count: usize, expand_bits: usize) {
        ffi(|| unsafe { sys::zkh_batch_expand_into_evaluate_ntt(self.ctx.0, output.raw, input.raw, count, expand_bits) });
    }

    fn batch_interpolate_ntt(&self, io: &Self::Buffer<Self::Elem>, count: usize) {
        ffi(|| unsafe { sys::zkh_batch_interpolate_ntt(self.ctx.0, io.raw, count) });
    }

    fn batch_bit_reverse(&self, io: &Self::Buffer<Self::Elem>, count: usize) {
        ffi(|| unsafe { sys::zkh_batch_bit_reverse(self.ctx.0, io.raw, count) });
    }

    fn batch_evaluate_any(&self, coeffs: &Self::Buffer<Self::Elem>, poly_count: usize, which: &Self::Buffer<u32>, xs: &Self::Buffer<Self::ExtElem>, out: &Self::Buffer<Self::ExtElem>) {
        ffi(|| unsafe { sys::zkh_batch_evaluate_any(self.ctx.0, coeffs.raw, poly_count, which.raw, xs.raw, out.raw) });
    }

    fn zk_shift(&self, io: &Self::Buffer<Self::Elem>, count: usize) {
        ffi(|| unsafe { sys::zkh_zk_shift(self.ctx.0, io.raw, count) });
    }

    fn mix_poly_coeffs(&self, output: &Self::Buffer<Self::ExtElem>, mix_start: &Self::ExtElem, mix: &Self::ExtElem, input: &Self::Buffer<Self::Elem>, combos: &Self::Buffer<u32>, input_size: usize, count: usize) {
        ffi(|| unsafe { sys::zkh_mix_poly_coeffs(self.ctx.0, output.raw, Self::ext_words(mix_start), Self::ext_words(mix), input.raw, combos.raw, input_size, count) });
    }

    fn eltwise_add_elem(&self, output: &Self::Buffer<Self::Elem>, input1: &Self::Buffer<Self::Elem>, input2: &Self::Buffer<Self::Elem>) {
        ffi(|| unsafe { sys::zkh_eltwise_add_elem(self.ctx.0, output.raw, input1.raw, input2.raw) });
    }

    fn eltwise_sum_extelem(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::ExtElem>) {
        ffi(|| unsafe { sys::zkh_eltwise_sum_extelem(self.ctx.0, output.raw, input.raw) });
    }

    fn eltwise_copy_elem(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::Elem>) {
        ffi(|| unsafe { sys::zkh_eltwise_copy_elem(self.ctx.0, output.raw, input.raw) });
    }

    fn eltwise_zeroize_elem(&self, elems: &Self::Buffer<Self::Elem>) {
        ffi(|| unsafe { sys::zkh_eltwise_zeroize_elem(self.ctx.0, elems.raw) });
    }

    fn fri_fold(&self, output: &Self::Buffer<Self::Elem>, input: &Self::Buffer<Self::Elem>, mix: &Self::ExtElem) {
        ffi(|| unsafe { sys::zkh_fri_fold(self.ctx.0, output.raw, input.raw, Self::ext_words(mix)) });
    }

    fn hash_rows(&self, output: &Self::Buffer<Digest>, matrix: &Self::Buffer<Self::Elem>) {
        ffi(|| unsafe { sys::zkh_hash_rows(self.ctx.0, output.raw, matrix.raw) });
    }

    fn hash_fold(&self, io: &Self::Buffer<Digest>, input_size: usize, output_size: usize) {
        ffi(|| unsafe { sys::zkh_hash_fold(self.ctx.0, io.raw, input_size, output_size) });
    }

    fn gather_sample(&self, dst: &Self::Buffer<Self::Elem>, src: &Self::Buffer<Self::Elem>, idx: usize, size: usize, stride: usize) {
        ffi(|| unsafe { sys::zkh_gather_sample(self.ctx.0, dst.raw, src.raw, idx, size, stride) });
    }

    fn scatter(&self, into: &Self::Buffer<Self::Elem>, index: &[u32], offsets: &[u32], values: &[Self::Elem]) {
        let vals: &[u32] = bytemuck::cast_slice(values);
        ffi(|| unsafe { sys::zkh_scatter(self.ctx.0, into.raw, index.as_ptr(), offsets.as_ptr(), vals.as_ptr(), index.len(), vals.len()) });
    }

    fn prefix_products(&self, io: &Self::Buffer<Self::ExtElem>) {
        ffi(|| unsafe { sys::zkh_prefix_products(self.ctx.0, io.raw) });
    }

    fn combos_prepare(&self, combos: &Self::Buffer<Self::ExtElem>, coeff_u: &Self::Buffer<Self::ExtElem>, combo_count: usize, cycles: usize, reg_sizes: &Self::Buffer<u32>, reg_combo_ids: &Self::Buffer<u32>, mix: &Self::ExtElem) {
        ffi(|| unsafe { sys::zkh_combos_prepare_regs(self.ctx.0, combos.raw, coeff_u.raw, combo_count, cycles, reg_sizes.size(), reg_sizes.raw, reg_combo_ids.raw, Self::ext_words(mix)) });
    }

    fn combos_divide(&self, combos: &Self::Buffer<Self::ExtElem>, chunks: Vec<(usize, Vec<Self::ExtElem>)>, cycles: usize) {
        // upstream: for every (combo, points) divide combo `combo` by (x - p) for each p in turn; the remainders must vanish
        for (combo, points) in chunks {
            let pts: &[u32] = bytemuck::cast_slice(&points);
            let rem = self.alloc::<Self::ExtElem>("combos_rem", points.len(), true);
            ffi(|| unsafe { sys::zkh_combos_divide(self.ctx.0, combos.raw, combo, cycles, pts.as_ptr(), points.len(), rem.raw) });
            rem.view(|r| debug_assert!(r.iter().all(|x| *x == Self::ExtElem::ZERO), "combos_divide: nonzero remainder"));
        }
    }
}

/// `impl CircuitHal<HipHal>`: a circuit is DATA here — the TapSet + PolyExtStep list of the upstream circuit crate
/// (`src/zirgen/{taps.rs, poly_ext.rs}`) serialised into the desc blob `tools/import_upstream_circuit.py` writes, loaded once;
/// its straight-line `eval_check` kernels are attached as code objects (`python -m zeth_amd.circuits.jit circuit.desc outdir/`).
pub struct HipCircuitHal {
    hal: Rc<HipHal>,
    circuit: *mut ZkhCircuit,
    kernels: RefCell<usize>,
}

impl HipCircuitHal {
    pub fn new(hal: Rc<HipHal>, desc: &[u32]) -> Self {
        let mut circuit: *mut ZkhCircuit = std::ptr::null_mut();
        ffi(|| unsafe { sys::zkh_circuit_load(hal.ctx.0, desc.as_ptr(), desc.len(), &mut circuit) });
        HipCircuitHal { hal, circuit, kernels: RefCell::new(0) }
    }

    /// One generated kernel (`.hsaco` image + its entry point) of `n_parts`; without any the on-device step interpreter runs.
    pub fn attach_code_object(&self, image: &[u8], kernel_name: &str, part: usize, n_parts: usize) {
        let name = CString::new(kernel_name).unwrap();
        ffi(|| unsafe { sys::zkh_circuit_attach_code_object_part(self.circuit, image.as_ptr() as *const _, image.len(), name.as_ptr(), part, n_parts) });
        *self.kernels.borrow_mut() = unsafe { sys::zkh_circuit_compiled_parts(self.circuit) };
    }
}

impl Drop for HipCircuitHal {
    fn drop(&mut self) {
        unsafe { sys::zkh_circuit_destroy(self.circuit) }
    }
}

impl CircuitHal<HipHal> for HipCircuitHal {
    fn eval_check(&self, check: &HipBuffer<BabyBearElem>, groups: &[&HipBuffer<BabyBearElem>], globals: &[&HipBuffer<BabyBearElem>], poly_mix: BabyBearExtElem, po2: usize, steps: usize) {
        let g: Vec<*const ZkhBuf> = groups.iter().map(|b| b.raw as *const ZkhBuf).collect();
        let gl: Vec<*const ZkhBuf> = globals.iter().map(|b| b.raw as *const ZkhBuf).collect();
        debug_assert_eq!(check.size(), 4 * INV_RATE * steps); // CHECK_SIZE planes of 4n words: 4 x 4n
        ffi(|| unsafe { sys::zkh_eval_check(self.hal.ctx.0, self.circuit, check.raw, g.as_ptr(), g.len(), gl.as_ptr(), gl.len(), HipHal::ext_words(&poly_mix), po2, steps, 0) });
    }
}

/// `SegmentProver::prove` in the two halves upstream drives `Prover` in, for hosts that would rather hand the library whole
/// traces than go op by op: `zkh_prove_begin` (header, commit code + data, draw the accum mix) -> the circuit's own
/// `accumulate` -> `zkh_prove_finish` (commit accum, eval_check, DEEP, FRI, queries).  Traces come from pinned memory
/// (`zkh_host_alloc`) through `zkh_write_async`: enqueued on the stream, no host sync.
pub fn prove_segment_from_host_traces(hal: &HipHal, prover: *mut sys::ZkhProver, po2: usize, code: &[u32], data: &[u32], out_global: &[u32], mix_words: usize,
                                      accumulate: impl FnOnce(&[u32], &HipBuffer<BabyBearElem>) -> HipBuffer<BabyBearElem>) -> Vec<u32> {
    let dcode = hal.alloc::<BabyBearElem>("code", code.len(), false);
    let ddata = hal.alloc::<BabyBearElem>("data", data.len(), false);
    ffi(|| unsafe { sys::zkh_write_async(hal.ctx.0, dcode.raw, code.as_ptr(), 0, code.len()) });
    ffi(|| unsafe { sys::zkh_write_async(hal.ctx.0, ddata.raw, data.as_ptr(), 0, data.len()) });
    let mut job: *mut sys::ZkhSealJob = std::ptr::null_mut();
    let mut mix = vec![0u32; mix_words.max(1)];
    ffi(|| unsafe { sys::zkh_prove_begin(prover, po2, dcode.raw, ddata.raw, out_global.as_ptr(), &mut job, mix.as_mut_ptr()) });
    let accum = accumulate(&mix[..mix_words], &ddata);
    let (mut seal, mut words): (*mut u32, usize) = (std::ptr::null_mut(), 0);
    ffi(|| unsafe { sys::zkh_prove_finish(job, accum.raw, &mut seal, &mut words) });
    let out = unsafe { std::slice::from_raw_parts(seal, words) }.to_vec();
    unsafe { sys::zkh_free_seal(seal) };
    out
}

/// `ProverServer::prove_session` + lift / join to ONE succinct receipt (risc0-zkvm 3.0.3 host/server/prove/prover_impl.rs), as the
/// library's session executor: `zkh_session_create` (G devices x K lanes), `zkh_session_build_recursion` (the lift / lift2 / join /
/// join3 — and, with assumptions, union / resolve — programs of a block with these segment sizes, BUILT by the library: no `.zkr` files), `zkh_session_prove(join_tree = 2)`
/// (seals and fold as one pipeline), and the root receipt's seal back.  `desc` = the segment circuit (`zkh_shipped_circuit_desc`
/// or the imported upstream tables), `segments` = (po2, seed) per segment, largest sizes first.
///
/// `chained` = Some((initial state, journal)) for a SYN-S circuit: the executor's pass gives every segment its pre-state and exit code, and the
/// LAST seal binds Output{SHA-256(journal), assumptions} — `journal` = the bytes the guest commits, for zeth the 32-byte block hash
/// (/root/reference/guests/stateless-client/src/lib.rs:33), which `cli.rs:103-107` then compares with `block.hash_slow()`.
pub fn prove_session_succinct(devices: &[i32], lanes_per_device: usize, desc: &[u32], segments: &[(u32, u64)],
                              assumptions: Option<(&[u32], &[AssumptionReceipt])>, chained: Option<(u32, &[u8])>) -> Vec<u32> {
    let mut session: *mut sys::ZkhSession = std::ptr::null_mut();
    ffi(|| unsafe { sys::zkh_session_create(devices.as_ptr(), devices.len(), lanes_per_device, desc.as_ptr(), desc.len(), std::ptr::null(), 0, &mut session) });
    if let Some((initial_state, journal)) = chained {
        ffi(|| unsafe { sys::zkh_session_set_chained(session, 1, initial_state) });
        ffi(|| unsafe { sys::zkh_session_set_journal(session, journal.as_ptr(), journal.len()) });
    }
    // `ProverServer::{union, resolve}`: the session's assumption receipts (keccak batches proven by `prove_keccak`) are handed over
    // BEFORE the programs are built; the executor lifts them, unites them pairwise and resolves the session's root against the union
    if let Some((adesc, receipts)) = assumptions {
        let seals: Vec<*const u32> = receipts.iter().map(|r| r.seal.as_ptr()).collect();
        let words: Vec<usize> = receipts.iter().map(|r| r.seal.len()).collect();
        let po2s: Vec<u32> = receipts.iter().map(|r| r.po2).collect();
        let roots: Vec<u32> = receipts.iter().flat_map(|r| r.control_root.iter().copied()).collect();
        ffi(|| unsafe { sys::zkh_session_set_assumptions(session, adesc.as_ptr(), adesc.len(), seals.as_ptr(), words.as_ptr(), po2s.as_ptr(), roots.as_ptr(), receipts.len()) });
    }
    let mut sizes: Vec<u32> = segments.iter().map(|s| s.0).collect();
    sizes.sort_unstable_by(|a, b| b.cmp(a));
    sizes.dedup();
    ffi(|| unsafe { sys::zkh_session_build_recursion(session, sizes.as_ptr(), sizes.len(), 1) });
    let segs: Vec<sys::ZkhSegment> = segments.iter().map(|&(po2, seed)| sys::ZkhSegment { po2, seed, ..unsafe { std::mem::zeroed() } }).collect();
    let mut info: sys::ZkhProveInfo = unsafe { std::mem::zeroed() };
    ffi(|| unsafe { sys::zkh_session_prove(session, segs.as_ptr(), segs.len(), 2, 0, std::ptr::null() /* a fresh OS key per fold proof */, &mut info) });
    ffi(|| unsafe { sys::zkh_session_verify(session, segs.as_ptr(), &info, 0) });
    let root = unsafe { std::slice::from_raw_parts(info.root_seal, info.root_seal_words) }.to_vec();
    unsafe { sys::zkh_prove_info_free(&mut info); sys::zkh_session_destroy(session) };
    root
}

/// One assumption receipt of a session: a seal of another circuit (KECCAK-F), its size, that circuit's control root at the size.
pub struct AssumptionReceipt { pub seal: Vec<u32>, pub po2: u32, pub control_root: [u32; 8] }

/// `verify_succinct` for a RESOLVED receipt: the claim is recomputed from the segment leaves AND the assumption receipts' claim digests
/// (the union tree sorts every pair; `leaves` empty: the receipt is the union-tree root alone).
pub fn verify_succinct_resolved(root_seal: &[u32], allowed_roots: &[[u32; 8]], root_program: usize, leaves: &[([u32; 8], u32, u32)], ranks: usize,
                                assumption_claims: &[[u32; 8]]) -> Result<(), String> {
    let flat_roots: Vec<u32> = allowed_roots.iter().flatten().copied().collect();
    let mut flat_leaves: Vec<u32> = Vec::with_capacity(10 * leaves.len());
    for (core, pre, post) in leaves { flat_leaves.extend_from_slice(core); flat_leaves.push(*pre); flat_leaves.push(*post); }
    let flat_claims: Vec<u32> = assumption_claims.iter().flatten().copied().collect();
    sys::ffi_wrap(|| unsafe {
        sys::zkh_succinct_verify_resolved(root_seal.as_ptr(), root_seal.len(), flat_roots.as_ptr(), allowed_roots.len(), root_program,
                                          flat_leaves.as_ptr(), leaves.len(), ranks, flat_claims.as_ptr(), assumption_claims.len())
    })
}

/// `SuccinctReceipt::verify_integrity` on the verifier's host (no GPU): one RECURSION seal under an allowed program's control
/// root, the allowed-programs root it carries, and its claim = the root of the leaves' claim tree (`leaves`: claim digest, pre, post).
pub fn verify_succinct(root_seal: &[u32], allowed_roots: &[[u32; 8]], root_program: usize, leaves: &[([u32; 8], u32, u32)], ranks: usize) -> Result<(), String> {
    let flat_roots: Vec<u32> = allowed_roots.iter().flatten().copied().collect();
    let mut flat_leaves: Vec<u32> = Vec::with_capacity(10 * leaves.len());
    for (core, pre, post) in leaves { flat_leaves.extend_from_slice(core); flat_leaves.push(*pre); flat_leaves.push(*post); }
    sys::ffi_wrap(|| unsafe {
        sys::zkh_succinct_verify(root_seal.as_ptr(), root_seal.len(), flat_roots.as_ptr(), allowed_roots.len(), root_program,
                                 flat_leaves.as_ptr(), leaves.len(), ranks)
    })
}
