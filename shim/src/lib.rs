//! Safe wrappers over `libwhisper_gfx950.so` for the three whisper servers.
//!
//! What each wrapper stands in for (paths relative to the reference tree):
//!
//! | here | replaces |
//! |---|---|
//! | [`Prio3Gpu::prepare_batched_many`] | the map over `prepare_encoded_key` in `run_vdaf_prepare{,_rayon}` (`server-batch-prio3/src/main.rs:75-91, 138-156`) |
//! | [`Prio3Gpu::prepare_batched`] | `BatchAggregator::prepare_batched` for one key (`batched-flpcp-rs/src/vdaf.rs:508-520`, impl `vdaf/prio3.rs:2466-2879`) as a batch of one |
//! | [`Prio3Gpu::sort_by_testing_id`] | `tags_and_shares.sort_by(testing_id)` (`main.rs:93, 158`) |
//! | [`Prio3Gpu::aggregate`] | the merge loop over good reports (`main.rs:108-118, 173-183`) |
//! | [`Prio3Gpu::allgather_reduce`] | the merge of the partial aggregates (`main.rs:327-335`) across the GPUs of one server |
//! | [`Prio3Gpu::prepare_init_many`] / [`Prio3Gpu::prepare_finish_many`] | both maps of `server-base-prio3/src/main.rs:41-100` |
//! | [`HhGpu::eval_and_check_many`] | the expansion `par_iter` of `server-hh/src/main.rs:163-174, 255-265` (`hhcore/src/protocol.rs:78`) |
//! | [`HhGpu::aggregate_hhs`] | the rayon `reduce` of `aggregate_hhs` (`server-hh/src/main.rs:72-104`) |
//!
//! Threading: a context may be shared (`Prio3Gpu: Send + Sync`); every call takes the context's recursive lock inside the
//! library.  A sequence that belongs together -- prepare, the host's group test, aggregate of THAT batch -- is bracketed with
//! [`Prio3Gpu::lock`], whose guard unlocks on drop; the 32 tokio tasks of `single_tag = false` queue behind it.
//!
//! This crate cannot be compiled in the image it was written in (no cargo / rustc there); `ffi.rs` is checked against the C
//! header by `tests/test_shim_matches_header.py`, and the same call sequences run in C++ (`tests/cpp/`) and Python (`tests/`).
pub mod ffi;

use std::ffi::CStr;
use std::fmt;
use std::ptr;

pub use ffi::{WhPrio3Params, WhPrio3Sizes, WH_AVERAGE, WH_HISTOGRAM, WH_SUMVEC};

/// A library-level failure (negative status of an entry point) with the context's last error text.
#[derive(Debug, Clone)]
pub struct WhError {
    pub code: i32,
    pub what: String,
}

impl fmt::Display for WhError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "whisper_gfx950 error {}: {}", self.code, self.what)
    }
}
impl std::error::Error for WhError {}

fn last_error(ctx: *const ffi::WhCtx) -> String {
    // valid until the next call on the context: copied at once
    unsafe {
        let p = ffi::wh_last_error(ctx);
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

fn check(ctx: *const ffi::WhCtx, rc: i32, call: &str) -> Result<(), WhError> {
    if rc == ffi::WH_OK {
        Ok(())
    } else {
        Err(WhError { code: rc, what: format!("{call}: {}", last_error(ctx)) })
    }
}

/// `Vec<Vec<u8>>` (what `ClientsPool::subscribe_and_get` hands the servers, `main.rs:268-274`) as one blob + offsets.
pub fn flatten(keys: &[Vec<u8>]) -> (Vec<u8>, Vec<u64>) {
    let total: usize = keys.iter().map(|k| k.len()).sum();
    let mut blob = Vec::with_capacity(total + 1);
    let mut offsets = Vec::with_capacity(keys.len() + 1);
    offsets.push(0u64);
    for k in keys {
        blob.extend_from_slice(k);
        offsets.push(blob.len() as u64);
    }
    blob.push(0); // the library never reads it; keeps `as_ptr()` of an empty batch valid
    (blob, offsets)
}

/// Output of one batched verification: per report the proof tag (already negated for `agg_id != 0`, as
/// `prepare_encoded_key` returns it, `main.rs:53-59`), the testing id and the failure class (0 = verified).
pub struct PreparedBatch {
    pub tags: Vec<[u8; 16]>,
    pub testing_ids: Vec<u128>,
    pub status: Vec<i32>,
}

/// One Prio3 instance on one GPU (`Prio3::new_sum_vec_256(2, 16, vec_size, chunk_size)` + `Prf::new(verify_key)`).
pub struct Prio3Gpu {
    ctx: *mut ffi::WhCtx,
    pub params: WhPrio3Params,
    pub sizes: WhPrio3Sizes,
}

// the library serialises callers on the context's own lock
unsafe impl Send for Prio3Gpu {}
unsafe impl Sync for Prio3Gpu {}

/// Holds the context's lock; other threads' calls wait until it is dropped.
pub struct CtxGuard<'a> {
    ctx: *mut ffi::WhCtx,
    _owner: std::marker::PhantomData<&'a ()>,
}
impl Drop for CtxGuard<'_> {
    fn drop(&mut self) {
        unsafe { ffi::wh_ctx_unlock(self.ctx) };
    }
}

impl Prio3Gpu {
    pub fn new(device: i32, params: WhPrio3Params) -> Result<Self, WhError> {
        let mut sizes = WhPrio3Sizes::default();
        let rc = unsafe { ffi::wh_prio3_sizes_of(&params, &mut sizes) };
        if rc != ffi::WH_OK {
            return Err(WhError { code: rc, what: "invalid Prio3 parameters".into() });
        }
        let mut ctx: *mut ffi::WhCtx = ptr::null_mut();
        let rc = unsafe { ffi::wh_ctx_create(device, &params, &mut ctx) };
        if rc != ffi::WH_OK {
            return Err(WhError { code: rc, what: "wh_ctx_create (no GPU? there is no CPU fallback)".into() });
        }
        Ok(Prio3Gpu { ctx, params, sizes })
    }

    /// `Prio3::new_sum_vec_256(num_aggregators, bits, len, chunk_length)` (prio3.rs:150-165)
    pub fn new_sum_vec_256(device: i32, num_aggregators: u32, bits: u32, len: u32, chunk: u32) -> Result<Self, WhError> {
        Self::new(device, WhPrio3Params { circuit: WH_SUMVEC, bits, len, chunk, num_aggregators, alg_id: 0 })
    }
    /// `Prio3::new_histogram_256(num_aggregators, length, chunk_length)` (prio3.rs:300-313)
    pub fn new_histogram_256(device: i32, num_aggregators: u32, len: u32, chunk: u32) -> Result<Self, WhError> {
        Self::new(device, WhPrio3Params { circuit: WH_HISTOGRAM, bits: 0, len, chunk, num_aggregators, alg_id: 0 })
    }
    /// `Prio3::new_average_256(num_aggregators, bits)` (prio3.rs:360-380); the servers use `AVG_BITS = 64`
    pub fn new_average_256(device: i32, num_aggregators: u32, bits: u32) -> Result<Self, WhError> {
        Self::new(device, WhPrio3Params { circuit: WH_AVERAGE, bits, len: 0, chunk: 0, num_aggregators, alg_id: 0 })
    }

    pub fn raw(&self) -> *mut ffi::WhCtx {
        self.ctx
    }

    pub fn lock(&self) -> CtxGuard<'_> {
        unsafe { ffi::wh_ctx_lock(self.ctx) };
        CtxGuard { ctx: self.ctx, _owner: std::marker::PhantomData }
    }

    /// The per-report loop of `run_vdaf_prepare{,_rayon}` for a whole `Vec` of encoded `VdafBatchedKey`s.  A report the
    /// reference would `unwrap()`-abort on comes back with a non-zero status and a zero tag; the out shares stay in HBM
    /// until [`Prio3Gpu::aggregate`].
    pub fn prepare_batched_many(&self, verify_key: &[u8; 16], keys: &[Vec<u8>]) -> Result<PreparedBatch, WhError> {
        let n = keys.len();
        let (blob, offsets) = flatten(keys);
        let mut tags = vec![0u8; 16 * n];
        let mut ids = vec![0u8; 16 * n];
        let mut status = vec![0i32; n];
        let rc = unsafe {
            ffi::wh_prio3_prepare_batch(self.ctx, verify_key.as_ptr(), blob.as_ptr(), offsets.as_ptr(), n as u64,
                                        tags.as_mut_ptr(), ids.as_mut_ptr(), status.as_mut_ptr())
        };
        check(self.ctx, rc, "wh_prio3_prepare_batch")?;
        Ok(PreparedBatch {
            tags: tags.chunks_exact(16).map(|c| c.try_into().unwrap()).collect(),
            testing_ids: ids.chunks_exact(16).map(|c| u128::from_le_bytes(c.try_into().unwrap())).collect(),
            status,
        })
    }

    /// `prepare_encoded_key` for ONE encoded key (`main.rs:33-60`): tag (negated for `agg_id != 0`), testing id and the
    /// encoded out share.  The trait method `BatchAggregator::prepare_batched` is this with the Helper's tag negated
    /// back: `if key.agg_id != 0 { tag = -tag }` on the decoded `Field128`.  A batch of one does not fill a GPU: servers
    /// should call [`Prio3Gpu::prepare_batched_many`].
    pub fn prepare_batched(&self, verify_key: &[u8; 16], encoded_key: &[u8]) -> Result<([u8; 16], u128, Vec<u8>), WhError> {
        let _g = self.lock();
        let one = [encoded_key.to_vec()];
        let b = self.prepare_batched_many(verify_key, &one)?;
        if b.status[0] != 0 {
            return Err(WhError { code: b.status[0], what: "report rejected (decode / query error class)".into() });
        }
        let mut out = vec![0u8; 16 * self.sizes.output_len as usize];
        let rc = unsafe { ffi::wh_prio3_out_shares(self.ctx, out.as_mut_ptr()) };
        check(self.ctx, rc, "wh_prio3_out_shares")?;
        Ok((b.tags[0], b.testing_ids[0], out))
    }

    /// Stable sort permutation by testing id: position k of the sorted batch holds report `order[k]`.
    pub fn sort_by_testing_id(&self, testing_ids: &[u128]) -> Result<Vec<u32>, WhError> {
        let n = testing_ids.len();
        let mut ids = Vec::with_capacity(16 * n);
        for id in testing_ids {
            ids.extend_from_slice(&id.to_le_bytes());
        }
        let mut order = vec![0u32; n];
        let rc = unsafe { ffi::wh_sort_by_testing_id(self.ctx, ids.as_ptr(), n as u64, order.as_mut_ptr()) };
        check(self.ctx, rc, "wh_sort_by_testing_id")?;
        Ok(order)
    }

    /// Sum mod p of the out shares of the last batch over `keep[i] != 0` (None = all); reports with a non-zero status are
    /// never added.  Returns `output_len` encoded Field128 elements (`AggregateShare::get_decoded_with_param` reads them).
    pub fn aggregate(&self, keep: Option<&[u8]>) -> Result<Vec<u8>, WhError> {
        let mut agg = vec![0u8; 16 * self.sizes.output_len as usize];
        let rc = unsafe { ffi::wh_prio3_aggregate(self.ctx, keep.map_or(ptr::null(), |k| k.as_ptr()), agg.as_mut_ptr()) };
        check(self.ctx, rc, "wh_prio3_aggregate")?;
        Ok(agg)
    }

    /// Rank 0 of a multi-GPU server makes the id and hands it to the other ranks over the channel it already has.
    pub fn comm_unique_id() -> Result<[u8; ffi::WH_COMM_ID_BYTES], WhError> {
        let mut id = [0u8; ffi::WH_COMM_ID_BYTES];
        let rc = unsafe { ffi::wh_comm_unique_id(id.as_mut_ptr()) };
        if rc != ffi::WH_OK {
            let why = unsafe { CStr::from_ptr(ffi::wh_comm_why()).to_string_lossy().into_owned() };
            return Err(WhError { code: rc, what: format!("wh_comm_unique_id: {why}") });
        }
        Ok(id)
    }

    /// Collective: every rank calls it once.  `id = None` with `world = 1` is a lone GPU without RCCL.
    pub fn comm_init(&self, id: Option<&[u8; ffi::WH_COMM_ID_BYTES]>, rank: i32, world: i32) -> Result<(), WhError> {
        let rc = unsafe { ffi::wh_comm_init(self.ctx, id.map_or(ptr::null(), |i| i.as_ptr()), rank, world) };
        check(self.ctx, rc, "wh_comm_init")
    }

    /// This GPU's masked aggregate of the last batch, combined with every other rank's (RCCL all-gather of the raw limbs +
    /// on-GPU mod-p sum): the same bytes on every rank.
    pub fn allgather_reduce(&self, keep: Option<&[u8]>) -> Result<Vec<u8>, WhError> {
        let mut agg = vec![0u8; 16 * self.sizes.output_len as usize];
        let rc = unsafe { ffi::wh_prio3_allgather_reduce(self.ctx, keep.map_or(ptr::null(), |k| k.as_ptr()), agg.as_mut_ptr()) };
        check(self.ctx, rc, "wh_prio3_allgather_reduce")?;
        Ok(agg)
    }

    /// First map of `server-base-prio3`'s `run_vdaf_prepare`: encoded `VdafKey`s in, encoded `Prio3PrepareShare`s out
    /// (`(verifier_len + 1) * 16` bytes each) with the per-key status.
    pub fn prepare_init_many(&self, verify_key: &[u8; 16], keys: &[Vec<u8>]) -> Result<(Vec<u8>, Vec<i32>), WhError> {
        let n = keys.len();
        let (blob, offsets) = flatten(keys);
        let share_len = 16 * (self.sizes.verifier_len as usize + 1);
        let mut prep = vec![0u8; share_len * n];
        let mut status = vec![0i32; n];
        let rc = unsafe {
            ffi::wh_prio3_prepare_init_batch(self.ctx, verify_key.as_ptr(), blob.as_ptr(), offsets.as_ptr(), n as u64,
                                             prep.as_mut_ptr(), status.as_mut_ptr())
        };
        check(self.ctx, rc, "wh_prio3_prepare_init_batch")?;
        Ok((prep, status))
    }

    /// Second map: `prep_shares[a]` = aggregator a's prepare shares of the whole batch, in `agg_id` order.  Returns the
    /// prepare messages (the joint-rand seeds, 16 bytes each) and the final status per report.
    pub fn prepare_finish_many(&self, prep_shares: &[&[u8]], n: usize) -> Result<(Vec<u8>, Vec<i32>), WhError> {
        let ptrs: Vec<*const u8> = prep_shares.iter().map(|s| s.as_ptr()).collect();
        let mut msgs = vec![0u8; 16 * n];
        let mut status = vec![0i32; n];
        let rc = unsafe { ffi::wh_prio3_prepare_finish_batch(self.ctx, ptrs.as_ptr(), n as u64, msgs.as_mut_ptr(), status.as_mut_ptr()) };
        check(self.ctx, rc, "wh_prio3_prepare_finish_batch")?;
        Ok((msgs, status))
    }
}

impl Drop for Prio3Gpu {
    fn drop(&mut self) {
        unsafe { ffi::wh_ctx_destroy(self.ctx) };
    }
}

/// One `server-hh` process on one GPU.
pub struct HhGpu {
    ctx: *mut ffi::WhCtx,
}
unsafe impl Send for HhGpu {}
unsafe impl Sync for HhGpu {}

impl HhGpu {
    pub fn new(device: i32) -> Result<Self, WhError> {
        let mut ctx: *mut ffi::WhCtx = ptr::null_mut();
        let rc = unsafe { ffi::wh_hh_ctx_create(device, &mut ctx) };
        if rc != ffi::WH_OK {
            return Err(WhError { code: rc, what: "wh_hh_ctx_create".into() });
        }
        Ok(HhGpu { ctx })
    }

    /// Bytes of one bincode-serialised `KeyChain<u32>` for `NumBuckets = 2^depth`.
    pub fn key_chain_len(depth: u32) -> usize {
        unsafe { ffi::wh_hh_key_chain_len(depth) as usize }
    }

    /// `Prf::compute_prf` + `BCSketch::eval_and_check` for n key chains (`keys` = n x `key_chain_len(depth)` bincode bytes).
    /// Returns the 64-character proof strings, the testing ids and the status; the bucket vectors stay in HBM for
    /// [`HhGpu::aggregate_hhs`].
    pub fn eval_and_check_many(&self, verify_key: &[u8; 16], keys: &[u8], depth: u32) -> Result<(Vec<String>, Vec<u128>, Vec<i32>), WhError> {
        let stride = Self::key_chain_len(depth);
        assert!(stride > 0 && keys.len() % stride == 0, "keys must hold whole key chains of this depth");
        let n = keys.len() / stride;
        let mut tags = vec![0u8; 64 * n];
        let mut ids = vec![0u8; 16 * n];
        let mut status = vec![0i32; n];
        let rc = unsafe {
            ffi::wh_hh_eval_and_check_batch(self.ctx, verify_key.as_ptr(), keys.as_ptr(), stride as u64, n as u64, depth,
                                            tags.as_mut_ptr(), ids.as_mut_ptr(), status.as_mut_ptr())
        };
        check(self.ctx, rc, "wh_hh_eval_and_check_batch")?;
        Ok((
            tags.chunks_exact(64).map(|c| String::from_utf8_lossy(c).into_owned()).collect(),
            ids.chunks_exact(16).map(|c| u128::from_le_bytes(c.try_into().unwrap())).collect(),
            status,
        ))
    }

    /// `aggregate_hhs`' reduce over the kept clients of the last batch: `2^depth * 258` wrapping u32 sums
    /// (256 data counters, sign, ctr per bucket: `Bucket { data, sign_data, ctr }`).
    pub fn aggregate_hhs(&self, keep: Option<&[u8]>, depth: u32) -> Result<Vec<u32>, WhError> {
        let mut acc = vec![0u32; 258usize << depth];
        let rc = unsafe { ffi::wh_hh_aggregate(self.ctx, keep.map_or(ptr::null(), |k| k.as_ptr()), acc.as_mut_ptr()) };
        check(self.ctx, rc, "wh_hh_aggregate")?;
        Ok(acc)
    }
}

impl Drop for HhGpu {
    fn drop(&mut self) {
        unsafe { ffi::wh_ctx_destroy(self.ctx) };
    }
}
