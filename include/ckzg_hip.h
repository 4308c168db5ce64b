/*
 * ckzg_hip.h -- additive entry points of the MI355X build (not in the reference).
 *
 * The reference C API is one-blob-per-call except for the two verify_*_batch functions
 * (src/eip4844/eip4844.h:43-81, src/eip7594/eip7594.h:35-57); its only batched use is the Go
 * benchmark's goroutine fan-out (bindings/go/main_test.go:953-971).  A GPU wants the batch in one
 * call, and a caller that already holds blobs in HBM wants to pass device pointers.  These
 * symbols sit next to the unchanged ckzg.h ones; every one of them is plain C: pointers + sizes.
 *
 * "_device" variants take/return HIP device pointers on the GPU the settings were loaded on and
 * enqueue on the context's stream, returning after the work has completed.  Every pointer of such a call must be
 * device (or managed) memory of ONE GPU that holds tables of the KZGSettings: a host pointer, a pointer the HIP
 * runtime does not know, or buffers on different GPUs give C_KZG_BADARGS, nothing is launched.
 */
#ifndef CKZG_HIP_H
#define CKZG_HIP_H

#include "ckzg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Options of the NEXT load_trusted_setup* (snapshotted when the load starts).  Keys:
 *   "device"        HIP device ordinal (default: env CKZG_HIP_DEVICE, else LOCAL_RANK, else 0)
 *   "devices"       bit mask of devices to load on (bit i = device i, -1 = every visible device; env
 *                   CKZG_HIP_DEVICES = mask or "all").  0 (default): the single "device".  With several
 *                   devices every one holds its own tables; the host-pointer *_batch entry points and the
 *                   two verify_*_batch functions split their batch into contiguous ranges, one host thread
 *                   and one GPU per range, results written in place (the shape of the reference's goroutine
 *                   fan-out, bindings/go/main_test.go:953-971); single-unit calls go to whichever device
 *                   has a free stream.
 *   "streams"       concurrent calls per device (default 8): every call leases one slot = its own HIP
 *                   streams + scratch, the tables are shared and immutable, so threads that share a
 *                   KZGSettings (legal in the reference, bindings/rust/src/bindings/mod.rs:910-913) overlap
 *                   on the GPU; further callers wait for a free slot.
 *   "replicas"      independent table sets per selected device (default 1).  Test aid: 2 exercises the
 *                   multi-device fan-out on a one-GPU box.
 *   "commit_wbits"  window width c (4..16) of the fixed-base table over the 4096 Lagrange points
 *                   used by blob_to_kzg_commitment / compute_*_proof.  Tables cover 128-bit GLV
 *                   half-scalars: bytes = (floor(127/c)+1) * 4096 * 2^(c-1) * 96; default 10 -> 2.6 GB,
 *                   16 -> 103 GB
 *   "fk20_wbits"    window width of the FK20 fixed-base tables (8192 points); default: max(8, precompute)
 *   "proof_wbits"   window width of the table over the 4096 monomial points used by the low-latency
 *                   (no G1 FFT) cell-proof path; default 8 (0.8 GB), 0 disables the path
 *   "direct_max"    largest batch that takes the low-latency proof path; larger batches use FK20, which
 *                   does ~10x fewer point additions but costs ~4.3 ms for any small batch (4 dependent
 *                   ladder launches).  -1 (default): 1 blob, 2 blobs for a proof table of >= 13 bits, the
 *                   measured hand-over points; 0 disables the path
 *   "async_tables"  1: progressive widening.  load_trusted_setup builds the tables at the library's default widths
 *                   (10 / 8 / 8 bits: ~0.4 s) and returns a fully working KZGSettings; a background thread then builds
 *                   the requested wider tables one at a time (commitment, proof, FK20) and publishes each when it is
 *                   complete, so calls simply get faster while it runs (every call sees one consistent set of tables;
 *                   replaced tables stay allocated until free_trusted_setup).  ckzg_hip_wait_tables blocks until the
 *                   widening has finished, ckzg_hip_tables_ready polls it; free_trusted_setup cancels it.  0 (default):
 *                   the load builds the requested widths itself before it returns.  env CKZG_HIP_ASYNC_TABLES.
 *   "coalesce"      1 (default): threads that call blob_to_kzg_commitment, compute_cells_and_kzg_proofs,
 *                   compute_blob_kzg_proof, verify_blob_kzg_proof or recover_cells_and_kzg_proofs (same cell indices) concurrently on one
 *                   KZGSettings -- the reference API's only parallel shape, bindings/go/main_test.go:953-971 -- share
 *                   batch launches: a caller that finds fewer than "coalesce_active" launches of its operation in
 *                   flight runs its own one-unit call at once (a lone caller's latency is unchanged: no queue, no
 *                   timer); later arrivals copy their inputs into a page-locked batch buffer and are served together by
 *                   ONE launch of the batch path when a launch place frees up.  A unit the batch path rejects (a
 *                   non-canonical field element, an invalid commitment) fails its own caller only; single-blob
 *                   verifications share ONE batch verification, and when that does not come out true every member
 *                   runs its own afterwards (a bad proof never changes another caller's answer).  0: every call
 *                   leases a stream of its own, as in rounds 2-3.
 *   "coalesce_active"  launches of one operation in flight per device before callers start to queue (1..8, default 2:
 *                   one batch computes while the next is copied in and the previous is copied out)
 *   "gpu_sha_min"   smallest verify_blob_kzg_proof_batch size whose Fiat-Shamir challenges are hashed on
 *                   the GPU; 0 (default): automatic -- the host hashes (underneath the blob copy) unless this process's
 *                   share of the host cores ("host_threads") is too small for it: the estimated host time
 *                   n * 66 us / threads (320 us without the x86 SHA extensions) is compared with the blob copy plus the
 *                   GPU hash's ~4.9 ms.  Batches of at most 3 blobs always hash on the host.
 *                   Takes effect immediately (as does "host_threads"; every other option is read by load_trusted_setup).
 *   "verify_pipe_min"  smallest verify_blob_kzg_proof_batch (host pointers) that crosses PCIe in 256-blob chunks while
 *                   earlier chunks are already evaluated; default 1024.  Takes effect immediately.
 *   "verify_call_table"  1 (default): verifications of >= 8 blobs / >= 128 cells build a fixed-base table over the
 *                   points of the call and take their sums from it; 0: ladder sums (also what a call does by itself when
 *                   the device is too full for the table).  Takes effect immediately.
 *   "verify_cu_partition"  1 (default): ckzg_hip_verify_blob_kzg_proof_batch_device on 640 .. 8192 blobs runs its SHA-256
 *                   chain (one wave per 64 blobs, as fast as its SIMD issues for it) on a stream confined to a quarter
 *                   of the compute units and the point validation / call-time table, which run underneath it, on streams
 *                   confined to the rest (hipExtStreamCreateWithCUMask): 4096 blobs 7.5 -> 6.3 ms.  0: plain streams
 *                   (also what the call uses where the runtime refuses a masked stream).  Takes effect immediately.
 *                   (HIP offers no flags for a masked stream: unlike the library's other streams these three are not
 *                   hipStreamNonBlocking, i.e. they order themselves against work the process puts on the legacy NULL
 *                   stream while such a call runs.  A process that relies on NULL-stream work overlapping a large
 *                   resident verification sets the option to 0.)
 *   "commit_graph"  1 (default): a lone blob_to_kzg_commitment call submits its copies and kernels as ONE hipGraph (one
 *                   submission instead of six; -20 us).  The graph is built node by node (hipGraphAddMemcpyNode /
 *                   KernelNode), once per stream slot and table set -- no stream capture is involved, so HIP calls
 *                   made meanwhile by other threads of the process (this library's or anybody else's: RCCL, PyTorch)
 *                   cannot disturb it.  0: plain stream launches always.  2: diagnostic, build the graph anew on every
 *                   lone call.  Takes effect immediately.
 *   "host_threads"  host threads one process of this library may keep busy per call (challenge hashing, staging copies,
 *                   point decompression at load).  0 (default): the CPUs of the process's affinity mask divided by the
 *                   processes that share the host (LOCAL_WORLD_SIZE or the MPI / PMI / Slurm node-local rank counts; else
 *                   WORLD_SIZE clamped to 8, the GPUs of one node; 1 otherwise).  The helper pools are sized by it when they start (first use).
 *   "wait_deadline_ms"  longest time any wait inside the library may last, in milliseconds (default 30000; env
 *                   CKZG_HIP_WAIT_DEADLINE_MS).  The reference never waits (src/eip4844/eip4844.c:264-280 is straight-line
 *                   code); this library waits for the GPU, for a free stream slot and -- coalesced callers -- for the
 *                   launch another caller runs.  None of these waits is unbounded: past the deadline the call that waited
 *                   returns C_KZG_ERROR (src/common/ret.h:24-29: an internal failure, returned) and one line on stderr
 *                   names what it waited for.  A DEVICE wait that expires marks the device as not answering: its kernels
 *                   may still be running, so later calls on it fail at once with C_KZG_ERROR instead of queueing behind
 *                   them, and free_trusted_setup leaves the device state in place.  Takes effect immediately.
 * A width that does not fit the free HBM is narrowed at load time (ckzg_hip_table_wbits reports the result).
 * Returns C_KZG_BADARGS for an unknown key or out-of-range value. */
C_KZG_RET ckzg_hip_set_option(const char *key, int64_t value);

/* Host threads this process's helper pools are sized for (the "host_threads" option; automatic: CPUs of the affinity
 * mask / processes sharing the host, see ckzg_hip_set_option).  Needs no GPU. */
int ckzg_hip_host_thread_budget(void);

/* The graph of the lone one-blob blob_to_kzg_commitment call (option "commit_graph"), process-wide counts:
 * out[0] graphs built, out[1] builds that failed (the slot stays on plain stream launches), out[2] calls launched as a
 * graph.  Needs no GPU. */
void ckzg_hip_commit_graph_stats(uint64_t out[3]);

/* Number of visible HIP devices (0 if none / runtime missing). */
int ckzg_hip_device_count(void);

/* Number of table sets (devices x replicas) this KZGSettings was loaded on; 0 if it has no GPU state. */
int ckzg_hip_num_devices(const KZGSettings *s);

/* blob_to_kzg_commitment (src/eip4844/eip4844.c:264-280) over n blobs.  Host pointers.
 * out[i] is written for every blob whose field elements are all canonical; the call returns
 * C_KZG_BADARGS if any blob is not (exactly the blobs for which the one-blob call would), and
 * per-blob status is returned in status[i] (C_KZG_RET values) when status != NULL. */
C_KZG_RET ckzg_hip_blob_to_kzg_commitment_batch(KZGCommitment *out, uint8_t *status,
                                                const Blob *blobs, uint64_t n,
                                                const KZGSettings *s);

/* Same with blobs/out/status resident in HBM (device pointers); runs on the device that holds them and returns
 * when the results are in HBM.  Non-canonical blobs are reported through d_status ONLY (d_status[i] = 1 =
 * C_KZG_BADARGS): the return value does not reflect them -- deriving it would cost a copy back to the host on
 * every call -- and is C_KZG_OK unless the call itself failed.  The same holds for the cells + proofs form below. */
C_KZG_RET ckzg_hip_blob_to_kzg_commitment_batch_device(void *d_out48, void *d_status,
                                                       const void *d_blobs, uint64_t n,
                                                       const KZGSettings *s);

/* compute_cells_and_kzg_proofs (src/eip7594/eip7594.c:61-157) over n blobs; cells and/or proofs
 * may be NULL (not both).  cells: n*128 Cell, proofs: n*128 KZGProof. */
C_KZG_RET ckzg_hip_compute_cells_and_kzg_proofs_batch(Cell *cells, KZGProof *proofs,
                                                      uint8_t *status, const Blob *blobs,
                                                      uint64_t n, const KZGSettings *s);
C_KZG_RET ckzg_hip_compute_cells_and_kzg_proofs_batch_device(void *d_cells, void *d_proofs,
                                                             void *d_status, const void *d_blobs,
                                                             uint64_t n, const KZGSettings *s);

/* compute_blob_kzg_proof (src/eip4844/eip4844.c:496-535) over n blobs: proofs[i] opens blobs[i] at the
 * Fiat-Shamir challenge derived from (blobs[i], commitments[i]).  Returns C_KZG_BADARGS if any blob
 * or commitment is invalid (per-blob C_KZG_RET in status[i] when status != NULL). */
C_KZG_RET ckzg_hip_compute_blob_kzg_proof_batch(KZGProof *proofs, uint8_t *status, const Blob *blobs,
                                                const Bytes48 *commitments_bytes, uint64_t n,
                                                const KZGSettings *s);

/* verify_blob_kzg_proof_batch (src/eip4844/eip4844.c:775-844) with blobs, commitments and proofs resident in HBM
 * (device pointers on one GPU: n Blob, n Bytes48, n Bytes48).  Point validation, bytes -> field elements, the
 * Fiat-Shamir challenges (SHA-256 of every blob, on the GPU), the evaluations and the three random-linear-combination
 * sums run on the device; 96 + 64 bytes per blob travel to the host for the batch transcript (eip4844.c:597-680)
 * and the two-pairing check runs there.  *ok is a HOST bool.  n == 0 gives *ok = true as in the reference.
 * Returns C_KZG_BADARGS exactly when the host-pointer call would (non-canonical field element, invalid point).
 * ckzg_hip_last_kernel_ms(s, 3) afterwards reports the device time of the call (which = 0: validation + conversion
 * + challenges + evaluation, which = 2: the sums).
 * The host-pointer verify_blob_kzg_proof_batch itself pipelines batches of >= 1024 blobs: chunks are DMA'd in place
 * when the caller's blobs are page-locked (hipHostMalloc / hipHostRegister), through pinned staging otherwise,
 * while earlier chunks are converted and evaluated.
 * (verify_cell_kzg_proof_batch has no resident form: its transcript is ONE SHA-256 stream over every cell,
 * eip7594.c:390-482, which only a host core can hash at a useful rate, so the cells must visit the host anyway.) */
C_KZG_RET ckzg_hip_verify_blob_kzg_proof_batch_device(bool *ok, const void *d_blobs, const void *d_commitments,
                                                      const void *d_proofs, uint64_t n, const KZGSettings *s);

/* recover_cells_and_kzg_proofs (src/eip7594/eip7594.c:177-304) over num_blobs rows that all hold
 * the SAME num_cells columns (the PeerDAS reconstruction case: a node has columns cell_indices[] of
 * every blob in the block).  cells is [num_blobs][num_cells], outputs are [num_blobs][128]; either
 * output may be NULL.  The vanishing polynomial of the missing set is built once for the batch.
 * status[i] (optional) is C_KZG_BADARGS for a row with a non-canonical field element. */
C_KZG_RET ckzg_hip_recover_cells_and_kzg_proofs_batch(Cell *recovered_cells, KZGProof *recovered_proofs,
                                                      uint8_t *status, const uint64_t *cell_indices,
                                                      const Cell *cells, uint64_t num_cells,
                                                      uint64_t num_blobs, const KZGSettings *s);

/* g1_lincomb_fast (src/common/lincomb.c:65-123) on the GPU: out = sum_i coeffs[i] * p[i] over `len` points in
 * the reference's in-memory forms (g1_t Jacobian, fr_t Montgomery); the empty sum is the identity.  The
 * library's own verify_*_batch paths call the same kernels on points they have already validated; here every
 * point must lie in the prime-order subgroup (or be the identity), otherwise C_KZG_BADARGS -- the reference
 * function accepts any curve point.  algo: 0 = what the library's own calls use (ladders, form chosen by size),
 * 1 = GLV ladders, form by size, 2 = bucket accumulation (pippenger.hip: the reference's method; slower than the ladders
 * below ~10^5 terms, which is why the library's own calls do not take it), 3 = ladders with one lane per GLV
 * half-term, 4 = ladders with four lanes per half-term (g1_quad.hpp). */
C_KZG_RET ckzg_hip_g1_lincomb(g1_t *out, const g1_t *p, const fr_t *coeffs, uint64_t len, int algo,
                              const KZGSettings *s);

/* Timing hook for bench.py: elapsed milliseconds of the named kernel family inside the last
 * batch call, measured with hipEvents on the stream the kernels were launched on (for a call that was
 * processed in several chunks: the last chunk).
 * which: 0 = scalar recoding, 1 = fixed-base MSM accumulate (k_msm_accumulate; k_msm_small on the FK20 path),
 *        2 = reduce+compress, 3 = whole device section, 4 = the two G1 FFTs of the FK20 path.
 * Returns a negative value if unavailable. */
double ckzg_hip_last_kernel_ms(const KZGSettings *s, int which);

/* Where the wall clock of the load that produced `s` went, in milliseconds (first device of the load).  Fills at
 * most n entries and returns how many:  0 hex text -> bytes (load_trusted_setup_file only), 1 host decompression of
 * the setup points + pairing sanity check + roots of unity, 2 HIP initialisation / code-object load / streams,
 * 3 small tables + subgroup check of the setup points, 4 / 5 allocation / construction of the commitment table,
 * 6 the 64 G1 FFTs of x_ext_fft_columns, 7 / 8 allocation / construction of the FK20 table, 9 / 10 the same for
 * the proof (monomial) table, 11 the remaining slots + host mirror of x_ext_fft_columns. */
int ckzg_hip_load_times(const KZGSettings *s, double *ms, int n);

/* Progressive widening ("async_tables"): block until the background table builds of the load that produced `s` have
 * finished (returns at once for an ordinary load) / 1 if they have, 0 while they run. */
C_KZG_RET ckzg_hip_wait_tables(const KZGSettings *s);
int ckzg_hip_tables_ready(const KZGSettings *s);

/* Coalescing counters of one operation of `s` since it was loaded.  op: 0 blob_to_kzg_commitment, 1 / 2 / 3
 * compute_cells_and_kzg_proofs with cells only / proofs only / both, 4 compute_blob_kzg_proof,
 * 5 recover_cells_and_kzg_proofs, 6 verify_blob_kzg_proof.  Fills at most n of: calls, calls that ran alone (idle
 * path), batch launches, calls served by batch launches, units in the largest launch, microseconds spent inside batch
 * launches, calls a batch could not answer and that ran alone afterwards (verifications only), open batches that sat on
 * an idle device until a member's periodic look released them (the net under the queueing protocol: 0 unless the
 * protocol has a hole), calls that left with C_KZG_ERROR at the wait deadline.  Returns the number filled (0: coalescing off). */
int ckzg_hip_coalesce_stats(const KZGSettings *s, int op, uint64_t *out, int n);

/* What the library is waiting for right now, written to file descriptor fd: every thread inside one of the library's
 * waits (what for, since when), per loaded KZGSettings the free stream slots and the queue state of every coalesced
 * operation.  For a process that has stopped making progress: takes no lock it could wait for (what is held is
 * reported as held), may be called from any thread, also while other calls are stuck.  Needs no GPU. */
void ckzg_hip_debug_dump(int fd);

/* out[0] the wait deadline in ms, out[1] waits that hit it since the library was loaded, out[2] bit mask of devices
 * marked as not answering.  Fills at most n entries and returns how many.  Needs no GPU. */
int ckzg_hip_wait_stats(uint64_t *out, int n);

/* Bytes of HBM held by the context's tables. */
uint64_t ckzg_hip_table_bytes(const KZGSettings *s);

/* Window width actually built for table `which` (0 = commitment, 1 = FK20, 2 = proof/monomial;
 * 0 if that table is disabled): load_trusted_setup narrows a requested width that does not fit the
 * free HBM instead of failing. */
int ckzg_hip_table_wbits(const KZGSettings *s, int which);

#ifdef __cplusplus
}
#endif
#endif /* CKZG_HIP_H */
