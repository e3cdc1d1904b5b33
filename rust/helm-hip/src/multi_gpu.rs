//! Multi-GPU evaluation for a Rust host: one process per GPU, keys and wire tables replicated, the gates of a
//! packed launch - the level of reference src/circuit.rs:531 is the sharded unit - split over the ranks, and the
//! launch's output ciphertexts all-gathered over RCCL / xGMI INSIDE libhelm_hip.so (include/helm_comm.h): this
//! crate needs no RCCL binding of its own.
//!
//! The 128-byte unique id is drawn by rank 0 (`HipComm::unique_id`) and has to reach the other processes by
//! whatever control plane the host has (a file, a socket, MPI); `HipComm::new` is collective over them.
//!
//! NOT COMPILED in this repository's image (no rustc); tests/test_rust_ffi_drift.py checks every declaration
//! used here against include/*.h, tests/test_gpu_rccl_world1.py runs the same calls on the GPU.
use crate::check;
use helm_hip_sys as sys;

pub struct HipComm {
    pub(crate) raw: *mut sys::helm_comm,
}

impl HipComm {
    /// ncclGetUniqueId (rank 0).
    pub fn unique_id() -> [u8; sys::HELM_COMM_ID_BYTES] {
        let mut id = [0u8; sys::HELM_COMM_ID_BYTES];
        check(unsafe { sys::helm_comm_get_unique_id(id.as_mut_ptr()) });
        id
    }

    /// What `new` needs from THIS process alone (an RCCL library is bound, the device can be made current), without entering
    /// a collective.  `ncclCommInitRank` blocks until every rank has entered it and has no timeout: a host calls this on every
    /// rank first, agrees on the results over its control plane, and only then lets anybody call `new`.
    pub fn precheck(device_id: i32) -> Result<(), String> {
        if unsafe { sys::helm_comm_precheck(device_id) } == 0 {
            Ok(())
        } else {
            Err(unsafe { std::ffi::CStr::from_ptr(sys::helm_hip_last_error()) }.to_string_lossy().into_owned())
        }
    }

    /// Communicators for ranks that are THREADS of this process (one thread and one engine context per rank, on
    /// `device_ids[rank]`): the all-gather is device-to-device copies inside the library, no RCCL involved.  Every rank calls
    /// each collective from its own thread; a rank that fails should `abort_group` so that the others return instead of waiting.
    pub fn in_process_group(device_ids: &[i32], timeout_s: f64) -> Vec<HipComm> {
        let mut raw: Vec<*mut sys::helm_comm> = vec![std::ptr::null_mut(); device_ids.len()];
        check(unsafe { sys::helm_comm_create_in_process(device_ids.as_ptr(), device_ids.len() as i32, timeout_s, raw.as_mut_ptr()) });
        raw.into_iter().map(|raw| HipComm { raw }).collect()
    }

    pub fn abort_group(&self) {
        unsafe { sys::helm_comm_abort_group(self.raw) };
    }

    /// ncclCommInitRank on `device_id`; blocks until all `world` processes holding `id` have called it.
    pub fn new(device_id: i32, id: &[u8; sys::HELM_COMM_ID_BYTES], rank: i32, world: i32) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::helm_comm_create(device_id, id.as_ptr(), rank, world, &mut raw) });
        HipComm { raw }
    }

    /// (rank, world size) as RCCL reports them.
    pub fn rank_and_world(&self) -> (i32, i32) {
        let (mut r, mut w) = (0, 0);
        check(unsafe { sys::helm_comm_info(self.raw, &mut r, &mut w, std::ptr::null_mut(), std::ptr::null_mut()) });
        (r, w)
    }

    pub fn barrier(&self) {
        check(unsafe { sys::helm_comm_barrier(self.raw) });
    }

    /// Maximum of `v` over the ranks (e.g. the elapsed time of a run).
    pub fn max_over_ranks(&self, v: f64) -> f64 {
        let mut x = v;
        check(unsafe { sys::helm_comm_all_reduce_f64(self.raw, &mut x, 1) });
        x
    }
}

impl Drop for HipComm {
    fn drop(&mut self) {
        unsafe { sys::helm_comm_destroy(self.raw) };
    }
}

impl<'a> crate::HipGateCircuit<'a> {
    /// Shard every launch of more than `replicate_below` bootstraps over the ranks of `comm` from the next
    /// `evaluate_encrypted` on (launches one GPU absorbs in one wave of workgroups are computed on every rank
    /// instead: 256 = one bootstrap per CU is the usual value).  Every rank must hold the same circuit, keys and
    /// input ciphertexts and make the same calls; every rank ends with the wire table of a one-GPU evaluation.
    /// The communicator must outlive the circuit's evaluations.
    /// Call before `encrypt_inputs` (which packs the launches for `world` ranks and uploads the program).
    pub fn shard_over(&mut self, comm: &HipComm, replicate_below: i64) {
        self.comm = comm.raw;
        self.world = comm.rank_and_world().1;
        self.replicate_below = replicate_below;
    }

    /// Overlap a launch's exchange with the launches that do not read its outputs (helm_hip_program_run_sharded_comm,
    /// overlap = 1: second stream, ring of gather buffers and dependency events inside the engine).  Same wire table.
    pub fn set_exchange_overlap(&mut self, on: bool) {
        self.overlap = on;
    }
}

impl<'a> crate::lut::HipLutCircuit<'a> {
    /// LUT mode: every bootstrap batch of at least `min_batch` look-ups is sharded over the ranks of `comm`
    /// (helm_si_set_exchange_comm); `capacity_rows` rows per rank travel per all-gather.
    pub fn shard_over(&mut self, comm: &HipComm, min_batch: i64, capacity_rows: i64) {
        check(unsafe { sys::helm_si_set_exchange_comm(self.ctx, comm.raw, min_batch, capacity_rows) });
    }
}
