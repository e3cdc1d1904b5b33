/*
 * helm_hip.h — C ABI of the MI355X-native gate-bootstrap engine.
 *
 * This header is the drop-in boundary for HELM's hot path.  In the reference the
 * boundary is not an FFI but the set of `tfhe` calls issued per gate:
 *
 *   reference src/gates.rs:254-275     Gate::evaluate_encrypted ->
 *        tfhe::boolean::ServerKey::{and,nand,or,nor,xor,xnor,mux,not,trivial_encrypt}
 *   reference src/circuit.rs:524-543   GateCircuit::evaluate_encrypted: per level,
 *        gates.par_iter_mut() -> one synchronous ServerKey call per gate
 *   reference src/circuit.rs:799-872   (feature "gpu", precedent for a batched shape)
 *        discard_{and,...}_lwe_ciphertext_vector(out, in1, in2, &bsk, &ksk, idx)
 *
 * The replacement is level-batched and device-resident: the wire table of
 * circuit.rs:517-520 (HashMap<String, Arc<RwLock<Ciphertext>>>) lives in HBM and
 * one call evaluates every gate of a netlist level.  INTEGRATION.md shows the
 * Rust `extern "C"` block + `impl EvalCircuit` shim a HELM maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative helm_status on failure;
 *     helm_hip_last_error() returns the message of the calling thread's last
 *     failure.  Nothing throws or aborts across this ABI.
 *   - the caller owns all host buffers; the library owns device memory behind the
 *     opaque handles.  A context is bound to one device and one stream and is not
 *     re-entrant (one context per host thread, or external locking).
 *   - level / batch calls are stream-asynchronous; helm_hip_sync() waits.
 *   - there is NO CPU fallback: without a usable gfx950 device
 *     helm_hip_ctx_create() fails with HELM_ERR_NO_DEVICE.
 *
 * Data layouts (torus_bits = 32: all words are uint32_t, arithmetic mod 2^32)
 *   LWE ciphertext      n mask words then 1 body word            (n+1 words)
 *   wire table          n_wires rows of n+1 words                (row-major)
 *   bootstrapping key   standard (coefficient) domain, as tfhe's
 *                       LweBootstrapKey: [n][pbs_l][k+1][k+1][N]
 *                       entry [i][j][r][c][*] = polynomial c of the GLWE row that
 *                       multiplies digit j (weight 2^(32-pbs_logB*(j+1))) of input
 *                       polynomial r in GGSW(s_i).
 *   keyswitching key    [k*N][ks_l][n+1]; entry [t][j] = LWE_small(
 *                       s_big[t] * 2^(32-ks_logB*(j+1)) )
 *   "big" LWE           k*N mask words + body (sample-extracted GLWE)
 *   boolean encoding    true = 1<<29 (+1/8), false = 7<<29 (-1/8)
 *                       (reference src/circuit.rs:29,33); decrypt = phase < 2^31
 *                       (src/circuit.rs:948)
 */
#ifndef HELM_HIP_H
#define HELM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct helm_hip_ctx helm_hip_ctx;
typedef struct helm_hip_wires helm_hip_wires;
typedef struct helm_hip_program helm_hip_program;

typedef enum {
    HELM_OK = 0,
    HELM_ERR_INVALID = -1,    /* bad argument / unsupported parameter set */
    HELM_ERR_NO_DEVICE = -2,  /* no usable gfx950 device                   */
    HELM_ERR_HIP = -3,        /* a HIP runtime call failed                 */
    HELM_ERR_STATE = -4,      /* keys not loaded, handle from other ctx …  */
    HELM_ERR_OOM = -5
} helm_status;

/* Runtime crypto parameters.  Replaces the hard-coded parameter choices of
 * reference src/bin/helm.rs:141-146 (n=512,k=1,N=1024,l=3,logB=7,ks 8/2) and
 * helm.rs:241 (tfhe::boolean::gen_keys() -> DEFAULT_PARAMETERS). */
typedef struct {
    int32_t torus_bits;      /* 32 (64 reserved for LUT / arithmetic mode)   */
    int32_t n;               /* small LWE dimension                          */
    int32_t k;               /* GLWE dimension                               */
    int32_t N;               /* polynomial size: 512 or 1024                 */
    int32_t pbs_l;           /* bootstrap decomposition level count          */
    int32_t pbs_logB;        /* bootstrap decomposition base log             */
    int32_t ks_l;            /* keyswitch decomposition level count          */
    int32_t ks_logB;         /* keyswitch decomposition base log             */
    int32_t pbs_order;       /* 0 = bootstrap then keyswitch (wires under the
                                small key; tfhe boolean default)             */
    int32_t grouping_factor; /* 1 (multi-bit PBS reserved)                   */
} helm_hip_params;

/* Gate op-codes = discriminants of `enum GateType`, reference src/gates.rs:23-45,
 * in declaration order. */
typedef enum {
    HELM_GATE_AND = 0, HELM_GATE_DFF = 1, HELM_GATE_LUT = 2, HELM_GATE_MUX = 3,
    HELM_GATE_NAND = 4, HELM_GATE_NOR = 5, HELM_GATE_NOT = 6, HELM_GATE_OR = 7,
    HELM_GATE_XNOR = 8, HELM_GATE_XOR = 9, HELM_GATE_BUF = 10,
    HELM_GATE_CONST_ONE = 11, HELM_GATE_CONST_ZERO = 12,
    HELM_GATE_MULT = 13, HELM_GATE_ADD = 14, HELM_GATE_SUB = 15, HELM_GATE_DIV = 16,
    HELM_GATE_SHL = 17, HELM_GATE_SHR = 18, HELM_GATE_COPY = 19
} helm_gate_op;

const char *helm_hip_last_error(void);
/* Number of visible HIP devices (does not initialise a device context). */
int helm_hip_device_count(void);

/* One HIP runtime per process.  A hipStream_t or a device pointer is only valid in the copy of libamdhip64 that made it;
 * a process that maps two copies (a host framework's bundled one next to the ROCm installation's this library's RUNPATH
 * names) and passes a handle from one to the other aborts inside the runtime.  Returns the number of distinct libamdhip64
 * objects mapped into the calling process and, when paths != NULL, their real paths separated by '\n' (truncated to cap).
 * Every entry point where a handle of the caller's crosses this ABI - helm_hip_set_stream, helm_si_set_stream,
 * helm_hip_wires_device_ptr, helm_hip_program_run_sharded and helm_si_set_exchange with a callback,
 * helm_comm_create_with_transport - fails with HELM_ERR_STATE and both paths in helm_hip_last_error() when this is > 1.
 * INTEGRATION.md "One HIP runtime per process" says how a host makes it 1. */
int helm_hip_runtime_copies(char *paths, size_t cap);

/* -- context --------------------------------------------------------------- */
/* Replaces ServerKey construction (reference src/bin/helm.rs:241, 187-192). */
int helm_hip_ctx_create(int device_id, const helm_hip_params *params, helm_hip_ctx **out);
int helm_hip_ctx_destroy(helm_hip_ctx *ctx);
/* The parameter set the context was created with. */
int helm_hip_get_params(const helm_hip_ctx *ctx, helm_hip_params *out);
/* Run on an existing hipStream_t (e.g. the host framework's current stream, so that collectives
 * ordered on it see the engine's kernels).  NULL = HIP's null (legacy default) stream.  A fresh
 * context runs on a non-blocking stream of its own. */
int helm_hip_set_stream(helm_hip_ctx *ctx, void *hip_stream);
int helm_hip_sync(helm_hip_ctx *ctx);
/* Bootstraps of one full round of the dominant (lockstep) build on this device: 4 per compute unit.
 * A launch of a whole number of rounds leaves no partial round behind; the host's launch packing
 * (helm_host_pack_levels) sizes launches with it.  Negative on error. */
int64_t helm_hip_launch_quantum(const helm_hip_ctx *ctx);

/* What a launch of at most 1/4, 2/4, 3/4 and 4/4 of helm_hip_launch_quantum() bootstraps costs on this context, relative to
 * a full round (cost[3] = 1): the engine runs a different build of the blind-rotate kernel per width (k_pbs_wide: one
 * bootstrap per CU on four SIMDs, k_pbs_duo: two per CU on two SIMDs each, k_pbs_trio: three per CU on four waves each, full
 * lockstep rounds).  Measured on
 * MI355X (profiles/r04/microbench.jsonl); the host's launch packing (helm_host_pack_levels_costed) sizes launches that are
 * narrower than a round with it. */
int helm_hip_launch_costs(const helm_hip_ctx *ctx, double cost[4]);

/* The prime field the blind-rotate kernels of this context compute in, as its size class: 49 = the lazy field p = 5072^4 + 1
 * (2^49.2; N = 512 sets whose exact products fit below p/2: no recentring inside transforms, and the first two stages of
 * every forward transform on decomposition digits as plain multiplications by the short eighth roots of unity 5072^k), 51 =
 * the 51-bit field p = 6432^4 + 1 (recentred at block boundaries), 50 = the lazy field p = 5440^4 + 1 (2^49.6) of N = 1024
 * contexts: helm_hip_load_bootstrap_key takes it when B/2 x the largest l1-norm of a column of the LOADED key stays below
 * p/2 - an exact guarantee for that key and every input (a generated key under helm.rs:141-146's set does; the worst case
 * of the set does not, and a key that does not fit keeps 51), so the value may change when a key is loaded
 * (HELM_HIP_FIELD=51 in the environment keeps the 51-bit field).  Results do not depend on it (exact integer arithmetic
 * either way); reported by benchmarks because the operation count of the kernels does.  Negative on error. */
int helm_hip_field_bits(const helm_hip_ctx *ctx);
/* Number of leading stages of every forward transform on decomposition digits that run as plain multiplications by short
 * roots of unity (2 in both fields of this engine: one radix-4 butterfly of 10 operations per four values instead of two
 * stages of modular butterflies; 0 when built with -DHELM_SHORT_ROOT_STAGES=0).  For benchmarks that count the operations
 * the kernels execute; results do not depend on it. */
int helm_hip_short_root_stages(const helm_hip_ctx *ctx);

/* Debug build only (-DHELM_CHECK_BOUNDS: csrc/libhelm_hip_check.so, loaded through HELM_HIP_LIB): the kernels of the boolean
 * engine count every violation of the contracts their lazy modular arithmetic rests on - counts[0] a modular multiplication
 * fed |a| >= 2^53, [1] a recentring fed the same, [2] a butterfly sum or difference that left 2^53 (or a plain short-root stage
 * that left (-p/2, p/2)), [3] the lean inverse transform entered with |x| > p/2, [4] a lifted value outside 2^51 - since the
 * last reset.  selftest != 0 first runs a kernel that breaks contract 0 once.  The regular build returns HELM_ERR_STATE. */
int helm_hip_bound_violations(helm_hip_ctx *ctx, uint32_t counts[8], int reset, int selftest);

/* -- keys ------------------------------------------------------------------ */
/* Replaces convert_lwe_bootstrap_key / convert_lwe_keyswitch_key (reference
 * src/bin/helm.rs:187-192): standard-domain keys from the host are uploaded and
 * the BSK is converted on the device to the engine's NTT domain. */
int helm_hip_load_bootstrap_key(helm_hip_ctx *ctx, const uint32_t *bsk_std, size_t n_words);
int helm_hip_load_keyswitch_key(helm_hip_ctx *ctx, const uint32_t *ksk, size_t n_words);

/* -- device-resident wire table -------------------------------------------- */
/* Replaces the HashMap<String, Arc<RwLock<Ciphertext>>> of circuit.rs:517-520. */
int helm_hip_wires_alloc(helm_hip_ctx *ctx, int64_t n_wires, helm_hip_wires **out);
int helm_hip_wires_free(helm_hip_ctx *ctx, helm_hip_wires *w);
/* lwe_host: count rows of n+1 words; row r goes to / comes from wire idx[r].  An upload must not
 * name the same wire twice (rows are written concurrently): HELM_ERR_INVALID. */
int helm_hip_wires_upload(helm_hip_ctx *ctx, helm_hip_wires *w, const int32_t *idx,
                          const uint32_t *lwe_host, int64_t count);
int helm_hip_wires_download(helm_hip_ctx *ctx, helm_hip_wires *w, const int32_t *idx,
                            uint32_t *lwe_host, int64_t count);
/* ServerKey::trivial_encrypt(value) (circuit.rs:455-458): zero mask, body = encoding. */
int helm_hip_wires_set_trivial(helm_hip_ctx *ctx, helm_hip_wires *w, const int32_t *idx,
                               const uint8_t *value, int64_t count);
/* Ciphertext::clone of rows (circuit.rs:517-520, 535): dst wire dst_idx[r] <- src wire src_idx[r], on the device;
 * src and dst may be the same table when the two index sets do not overlap. */
int helm_hip_wires_copy(helm_hip_ctx *ctx, helm_hip_wires *src, const int32_t *src_idx, helm_hip_wires *dst,
                        const int32_t *dst_idx, int64_t count);
/* Raw device pointer of the table (for collectives driven by the host). */
int helm_hip_wires_device_ptr(helm_hip_ctx *ctx, helm_hip_wires *w, void **dev_ptr, int64_t *n_wires);

/* -- one netlist level ----------------------------------------------------- */
/* Replaces the body of the level loop, reference src/circuit.rs:531-541:
 * `count` independent gates; gate g reads wires in0[g], in1[g], in2[g] (-1 =
 * unused) and writes wire out[g].  Operand meaning follows gates.rs:255-274:
 * binary gates use (in0,in1); MUX is sel=in2 ? in0 : in1; NOT/BUF/DFF use in0.
 * No gate of the level may read a wire written in the same level (guaranteed by
 * Circuit::compute_levels, circuit.rs:174-239).  Index arrays are host memory. */
int helm_hip_eval_gate_level(helm_hip_ctx *ctx, helm_hip_wires *w, const int32_t *opcode,
                             const int32_t *in0, const int32_t *in1, const int32_t *in2,
                             const int32_t *out, int64_t count);

/* -- whole levelised netlist ("program") ------------------------------------ */
/* The level_map of circuit.rs:174-239 uploaded once: gates sorted by level,
 * level_offsets[0..n_levels] delimiting them.  Avoids re-sending index arrays
 * every level / every cycle. */
int helm_hip_program_create(helm_hip_ctx *ctx, const int32_t *opcode, const int32_t *in0,
                            const int32_t *in1, const int32_t *in2, const int32_t *out,
                            const int64_t *level_offsets, int64_t n_levels,
                            helm_hip_program **prog);
int helm_hip_program_destroy(helm_hip_ctx *ctx, helm_hip_program *prog);
/* Evaluate levels [level_begin, level_end) in order (GateCircuit::evaluate_encrypted,
 * circuit.rs:506-549). */
int helm_hip_program_run(helm_hip_ctx *ctx, helm_hip_program *prog, helm_hip_wires *w,
                         int64_t level_begin, int64_t level_end);
/* Multi-GPU: this rank evaluates the contiguous slice `rank` of `world` of the
 * level's gates and packs the slice's output ciphertexts into `staging`
 * (device memory, chunk_rows(level, world) rows of n+1 words, zero padded).
 * The level is cut by BOOTSTRAP WEIGHT, not by gate count (binary gate 1, MUX 2,
 * NOT / BUF / DFF / constants 0: SURVEY 8(d)'s count), so that every rank gets
 * the same number of bootstraps to within one gate whatever the mix of a level:
 * helm_hip_program_chunk_bounds() returns the cut (bounds[r] .. bounds[r+1] are
 * rank r's gates, world + 1 entries), chunk_rows() the largest chunk = the rows
 * of one rank's slot in the all-gather.  rayon balances the same loop
 * dynamically in the reference (circuit.rs:531).
 * After an all-gather of the staging buffers (RCCL, driven by the caller),
 * helm_hip_program_scatter_level() writes all `world` chunks into the wire
 * table.  Keys and the wire table are replicated on every rank.
 * The chunk job lists and scatter tables of a (rank, world) pair are planned and
 * uploaded once - by helm_hip_program_shard_prepare(), or by the first
 * run_level_shard() - so that the per-level calls only launch kernels. */
int helm_hip_program_shard_prepare(helm_hip_ctx *ctx, helm_hip_program *prog, int rank, int world);
int64_t helm_hip_program_chunk_rows(helm_hip_program *prog, int64_t level, int world);
int helm_hip_program_chunk_bounds(helm_hip_program *prog, int64_t level, int world, int64_t *bounds);
int helm_hip_program_run_level_shard(helm_hip_ctx *ctx, helm_hip_program *prog, helm_hip_wires *w,
                                     int64_t level, int rank, int world, void *staging_dev);
int helm_hip_program_scatter_level(helm_hip_ctx *ctx, helm_hip_program *prog, helm_hip_wires *w,
                                   int64_t level, int world, const void *gathered_dev);
/* The whole sharded pass in one call - what a Rust host runs per evaluation: for every launch in order either
 * helm_hip_program_run() (launches of at most replicate_below bootstraps: one wave of workgroups absorbs them, every rank
 * computes them; world = 1 without a callback: every launch) or run_level_shard -> fn -> scatter_level.  fn(user, stage_dev, gather_dev, rows_per_rank) must all-gather
 * rows_per_rank rows of n+1 words of stage_dev from every rank into gather_dev in rank order ON THE CONTEXT'S STREAM
 * (ncclAllGather over RCCL/xGMI with the stream given to helm_hip_set_stream) and return 0.  stage_dev holds capacity_rows
 * rows, gather_dev capacity_rows * world.  Every rank issues the same call; identical wire tables on every rank, identical
 * to helm_hip_program_run() on one GPU.  The level is the sharded unit (reference src/circuit.rs:531). */
typedef int (*helm_hip_exchange_fn)(void *user, void *stage_dev, void *gather_dev, int64_t rows_per_rank);
int helm_hip_program_run_sharded(helm_hip_ctx *ctx, helm_hip_program *prog, helm_hip_wires *w, int rank, int world,
                                 int64_t replicate_below, void *stage_dev, void *gather_dev, int64_t capacity_rows,
                                 helm_hip_exchange_fn fn, void *user);
/* The same pass with the collective INSIDE the library: every sharded launch is computed straight into this rank's slot
 * of a gather buffer the program owns, all-gathered in place with ncclAllGather on the context's stream through `comm`
 * (include/helm_comm.h: an RCCL communicator created by the library; rank and world are the communicator's) and scattered
 * into the replicated wire table.  No callback, no host framework: what a Rust host and bench.py call.  world = 1 is valid and still
 * sends every launch of more than replicate_below bootstraps through stage -> ncclAllGather -> scatter (the path's
 * single-GPU test).  With timing enabled the all-gathers are bracketed by events (helm_hip_timing.exchange_ms).
 * overlap != 0: the exchange of a launch overlaps the launches that do not need its outputs - the chunk is computed on
 * the context's stream into one of three gather buffers, its ncclAllGather and the scatter run on a second stream of the
 * context, and a later launch waits (an event) only for the scatter of the last launch that writes one of its inputs;
 * the dependency table comes from the program itself.  Same wire table, bit for bit.  Needs every wire written at most
 * once per pass (helm_hip_program_overlap_applies() == 1; otherwise the call falls back to the in-order exchange). */
struct helm_comm;
int helm_hip_program_run_sharded_comm(helm_hip_ctx *ctx, helm_hip_program *prog, helm_hip_wires *w, struct helm_comm *comm,
                                      int64_t replicate_below, int overlap);
/* 1 when the overlapped exchange applies to this program (no wire written twice per pass), 0 otherwise. */
int helm_hip_program_overlap_applies(helm_hip_program *prog);
/* Number of programmable bootstraps a level costs (binary gate 1, MUX 2, others 0). */
int64_t helm_hip_program_level_pbs(helm_hip_program *prog, int64_t level);

/* -- primitive batch ops (tests, LUT / integer layers) ---------------------- */
/* Host-buffer convenience forms; each is the batched equivalent of one tfhe
 * primitive.  lwe_in: count x (n+1).  test_vectors: n_tv x N (body polynomial of
 * the accumulator); tv_index[g] selects one per ciphertext.  out_big: count x (k*N+1). */
int helm_hip_pbs_batch(helm_hip_ctx *ctx, const uint32_t *lwe_in, const uint32_t *test_vectors,
                       int64_t n_tv, const int32_t *tv_index, uint32_t *out_big, int64_t count);
/* in_big: count x (k*N+1) -> out: count x (n+1) */
int helm_hip_keyswitch_batch(helm_hip_ctx *ctx, const uint32_t *in_big, uint32_t *out, int64_t count);
/* Forward-transform one polynomial of N uint32 (as signed) and return the
 * engine's NTT-domain words, then invert: out[t] == in[t] (round-trip self test
 * of the fp64 NTT; used by tests). */
int helm_hip_ntt_roundtrip(helm_hip_ctx *ctx, const uint32_t *poly_in, uint32_t *poly_out, int64_t count);

/* -- timing ----------------------------------------------------------------- */
typedef struct {
    double pbs_ms;      /* accumulated device time in the blind-rotate kernel */
    double ks_ms;       /* ... keyswitch kernel                               */
    double linear_ms;   /* ... NOT/BUF/const kernel                           */
    int64_t pbs_launches, pbs_count; /* launches, bootstraps                   */
    int64_t ks_launches, ks_count;
    /* the lockstep build of the blind-rotate kernel alone (the full rounds of every launch: the
     * dominant kernel of wide levels); also contained in pbs_ms */
    double pbs_main_ms;
    int64_t pbs_main_launches, pbs_main_count;
    /* in-library exchanges (helm_hip_program_run_sharded_comm): device time between the events around every
     * ncclAllGather - it includes the wait for the slowest rank's chunk - their number and this rank's bytes */
    double exchange_ms;
    int64_t exchange_count, exchange_bytes;
} helm_hip_timing;
/* When enabled, HIP events on the context's stream bracket each kernel launch
 * (adds a sync per get_timing call, not per launch). */
int helm_hip_timing_enable(helm_hip_ctx *ctx, int enable);
int helm_hip_get_timing(helm_hip_ctx *ctx, helm_hip_timing *out, int reset);
/* Shader clock the chip held during the blind rotation of the most recent k_pbs launch on this
 * device (workgroup 0: s_memtime ticks per 100 MHz s_memrealtime tick), and that rotation's
 * duration.  The roofline of bench.py quotes its peak at the nominal clock and reports this one
 * beside it.  Synchronises the stream. */
int helm_hip_get_clock(helm_hip_ctx *ctx, double *shader_ghz, double *blind_rotation_ms);

#ifdef __cplusplus
}
#endif
#endif /* HELM_HIP_H */
