/*
 * helm_comm.h — C ABI of the multi-GPU exchange: an RCCL communicator owned by the library
 * (or, helm_comm_create_with_transport, one over a transport the host brings).
 *
 * The reference has no multi-GPU code; its unit of parallelism is the level
 * (reference src/circuit.rs:531 `gates.par_iter_mut()` for gates mode, :1057 for LUT mode,
 * :1321 for arithmetic mode).  BASELINE.json's north star shards that unit: the gates of a level
 * (here: of a packed launch) are split over the GPUs of a node, keys and wire tables replicated,
 * and only the small output LWE ciphertexts are all-gathered over RCCL / xGMI.
 *
 * With this header the collective lives INSIDE libhelm_hip.so: a host in any language creates one
 * communicator per process (one process per GPU), hands it to helm_hip_program_run_sharded_comm()
 * (include/helm_hip.h) or helm_si_set_exchange_comm() (include/helm_shortint.h), and needs neither
 * a host framework nor its own RCCL binding for the data path.  The callback forms
 * (helm_hip_program_run_sharded, helm_si_set_exchange) stay for hosts that bring their own collective.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1: the copy the process has already loaded - e.g.
 * a host framework's - is reused, otherwise the loader's search path, otherwise /opt/rocm/lib), so the library
 * loads on machines without RCCL; helm_comm_create() is what fails there (HELM_ERR_STATE).
 *
 * Conventions as in helm_hip.h: 0 on success, negative helm_status otherwise, message through
 * helm_hip_last_error().  A communicator is bound to one device; collectives are issued on the
 * stream passed in (the engine contexts pass their own), never on a hidden one.
 */
#ifndef HELM_COMM_H
#define HELM_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct helm_comm helm_comm;

#define HELM_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

/* 1 when an RCCL library could be bound in this process, 0 otherwise (never touches a device). */
int helm_comm_available(void);
/* Everything helm_comm_create() needs from THIS process alone, checked without entering a collective: an RCCL library
 * is bound, `device_id` exists and can be made current.  ncclCommInitRank blocks until every rank of the world has
 * entered it and has no timeout, so a host must not let one rank fail locally while the others go in: every rank calls
 * this first, the ranks agree on the results over their control plane (bench.py / helm_amd.comm.Comm.agree: a
 * MIN-reduce), and only then does anybody call helm_comm_create().  0 when the rank is ready. */
int helm_comm_precheck(int device_id);
/* ncclGetUniqueId: rank 0 calls it and sends the 128 bytes to the other ranks by any means it has
 * (a file, a socket, MPI, a host framework's key-value store). */
int helm_comm_get_unique_id(uint8_t id[HELM_COMM_ID_BYTES]);
/* ncclCommInitRank on `device_id`: collective over the `world` processes that share `id`
 * (world = 1 is a valid communicator: every collective still goes through RCCL). */
int helm_comm_create(int device_id, const uint8_t id[HELM_COMM_ID_BYTES], int rank, int world, helm_comm **out);
/* A communicator over a transport the HOST brings (MPI, a socket ring, a host framework's process group) instead of
 * RCCL: every collective of this header - and with it helm_hip_program_run_sharded_comm() and
 * helm_si_set_exchange_comm() - then calls `all_gather(user, send_dev, recv_dev, bytes_per_rank, hip_stream)`, which
 * must gather bytes_per_rank bytes of every rank into recv_dev in rank order (send_dev may be recv_dev + rank *
 * bytes_per_rank), ordered behind the work already queued on hip_stream, and return 0; the result must be in place for
 * work queued on hip_stream afterwards.  No RCCL is needed or touched (helm_comm_info reports rccl_version 0).  Also
 * how the rank > 0 offsets of the sharded paths are tested where every rank shares one GPU (tests/test_gpu_two_ranks*.py). */
typedef int (*helm_comm_all_gather_fn)(void *user, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *hip_stream);
int helm_comm_create_with_transport(int device_id, int rank, int world, helm_comm_all_gather_fn all_gather, void *user,
                                    helm_comm **out);
/* Communicators for ranks that are THREADS of this process (a host that drives its GPUs - or several contexts on one GPU -
 * from one process, one thread and one engine context per rank): out[r] is rank r's communicator on device device_ids[r].
 * The all-gather is device-to-device copies between the ranks' buffers (peer copies across devices) between two barriers;
 * every rank must call each collective from its own thread.  No RCCL is needed or touched (rccl_version 0).  A rank that
 * fails, or does not arrive within timeout_s seconds (<= 0: 600), breaks the group: every other rank's collective then
 * returns an error instead of waiting; helm_comm_abort_group() breaks it from outside a collective.  Also how the sharded
 * paths are tested at world size 8 on a box that allows six GPU processes (tests/test_gpu_eight_ranks.py). */
int helm_comm_create_in_process(const int *device_ids, int world, double timeout_s, helm_comm **out);
int helm_comm_abort_group(helm_comm *comm);
/* ncclCommDestroy.  NULL is accepted. */
int helm_comm_destroy(helm_comm *comm);
/* What RCCL itself reports for the communicator (ncclCommUserRank / ncclCommCount / ncclCommCuDevice /
 * ncclGetVersion); any out pointer may be NULL. */
int helm_comm_info(const helm_comm *comm, int *rank, int *world, int *device, int *rccl_version);
/* Collectives issued so far through this communicator and the bytes this rank contributed to them. */
int helm_comm_stats(const helm_comm *comm, int64_t *collectives, int64_t *bytes_sent);

/* ncclAllGather of bytes_per_rank bytes from every rank into recv_dev (rank order) on hip_stream.
 * In place when send_dev == recv_dev + rank * bytes_per_rank. */
int helm_comm_all_gather(helm_comm *comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *hip_stream);
/* Host-side helpers for a host without another control plane: max / sum of one double over the ranks
 * (ncclAllReduce on a private stream, synchronous), and a barrier (an all-reduce of one word). */
int helm_comm_all_reduce_f64(helm_comm *comm, double *value, int op /* 0 = sum, 1 = max */);
int helm_comm_barrier(helm_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* HELM_COMM_H */
