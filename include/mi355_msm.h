/* mi355_msm.h -- C ABI of the MI355X (gfx950) multi-scalar-multiplication engine.
 *
 * This is the drop-in boundary for the ZPrize-2022 prize1-msm hot path: plain pointers and sizes,
 * no C++ or torch types.  Every entry point names the reference interface it replaces
 * (paths relative to /root/reference, abbreviations as in SURVEY.md):
 *
 *   SPK  = open-division/prize1-msm/prize1a-msm-gpu/6block/sppark
 *   P1A  = open-division/prize1-msm/prize1a-msm-gpu
 *   CMB  = P1A/combined-top-solutions/combined-msm
 *   ARK  = open-division/prize4-msm-wasm/snarkify/zprize-prize4-15ac8c55-arkworks-algebra
 *
 * Data layouts (identical to what the reference FFI passes):
 *   bases    arkworks `Affine` images: x, y as 6 x u64 little-endian Montgomery (R = 2^384) limbs,
 *            then a 1-byte infinity flag; element stride is passed explicitly (104 B for G1).
 *            The flag byte is authoritative for infinity, not the coordinates.
 *   scalars  32 B each, 4 x u64 little-endian, plain integers (`BigInteger256`).
 *   results  arkworks `Projective` images (Jacobian X, Y, Z; 3 x 48 B Montgomery; 3 x 96 B for G2), written NORMALISED:
 *            (x, y, 1), or (1, 1, 0) for the point at infinity -- so equal points are equal bytes.
 *
 * Errors: `RustError { int code; char *message; }` returned by value, exactly sppark's convention
 * (SPK util/rusterror.h:15-27): code 0 = success, otherwise a hipError_t value (or -1 for argument
 * errors) and a malloc'd message the caller frees (Rust side: SPK rust/src/lib.rs:17-25).  A message is
 * ALWAYS supplied on failure, because ROCm has no cudaGetErrorString for the Rust macro to fall back on.
 * No C++ exception crosses this boundary.
 *
 * There is no CPU fallback: every compute entry point needs a visible gfx950 device and fails with
 * hipErrorNoDevice otherwise.
 */
#ifndef MI355_MSM_H
#define MI355_MSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int code;
  char* message;
} RustError;

typedef struct mi355_msm_ctx mi355_msm_ctx;

enum {
  MI355_BLS12_377_G1 = 0, /* fq: ARKC bls12_377/src/fields/fq.rs:4, curve b = 1 */
  MI355_BLS12_381_G1 = 1, /* fq: ARKC bls12_381/src/fields/fq.rs:4, curve b = 4 */
  MI355_BLS12_377_G2 = 2, /* coordinates in Fq2 = Fq[u]/(u^2+5) (ARKC bls12_377/src/fields/fq2.rs:13, curves/g2.rs:47-78):
                             Affine images are 200 B (x.c0 x.c1 y.c0 y.c1, flag at byte 192), Projective images 288 B */
  MI355_BLS12_381_G2 = 3  /* Fq2 = Fq[u]/(u^2+1), b' = 4(1+u) (ARKC bls12_381/src/fields/fq2.rs:13, curves/g2.rs:47-48, 74-91);
                             same images as MI355_BLS12_377_G2 */
};

/* Stage indices of mi355_msm_last_timings(). */
enum {
  MI355_T_DIGITS = 0,
  MI355_T_SORT = 1,
  MI355_T_ACCUMULATE = 2, /* the dominant kernel (bucket accumulation) */
  MI355_T_SEGREDUCE = 3,
  MI355_T_BUCKET_REDUCE = 4,
  MI355_T_HOST_FOLD = 5,
  MI355_T_TOTAL = 6,
  MI355_T_COUNT = 8
};

/* ---- canonical context API ---------------------------------------------------------------------
 * Replaces the context half of the harness FFI: mult_pippenger_init / mult_pippenger_inf
 * (P1A 6block/cuda/pippenger_inf.cu:50-53, 87-92) and MSMAllocContext / MSMPreprocessPoints / MSMRun /
 * MSMFreeContext (CMB MSM.h:72-75).  Bases are uploaded and converted once; each run streams scalars. */

/* device < 0 selects the current HIP device. */
RustError mi355_msm_create(mi355_msm_ctx** out, int curve, int device);
RustError mi355_msm_destroy(mi355_msm_ctx* ctx);

/* ---- one MSM over several GPUs, behind the SAME context API ----------------------------------------
 * The reference is single-device (SPK msm/pippenger.cuh:400-416 hard-codes device 0), but its harness only ever calls
 * init + run (P1A 6block/src/lib.rs:54-109; CMB MSM.h:72-75), so sharding has to live behind those calls.  A sharded context
 * is an ordinary mi355_msm_ctx*: set_bases / set_bases_device / set_bases_serialized / run / run_device / set_option / query /
 * last_timings / destroy all accept it.  Shard g of G owns the contiguous slice [g*ceil(n/G), ...) of the bases and of every
 * scalar batch, one host thread + one device context + one stream per shard (the multi-stream orchestration of
 * P1A matter-labs/src/lib.rs:125-201 with devices in place of streams); each shard's host thread returns one folded partial
 * point per batch, so the G partials are already in host memory and the "final 8-point curve add" is a host fold
 * (mi355_msm_fold) -- no collective is run by default.  Option "combine" = 2 additionally sends the partials through one
 * ncclAllGather over RCCL/xGMI (single-process communicator, built at set_bases; librccl is dlopen'ed) and requires the
 * exchanged copy to equal what was sent: a link check for bring-up, not a step the result depends on (0 / 1: host fold only).
 * The collective that IS needed -- partials in different processes -- is the one-process-per-GPU path (dist.py, torchrun).
 * Queries: "shards", "rccl_exchanges"; counters of the single-device queries add up over the shards. */
RustError mi355_msm_create_sharded(mi355_msm_ctx** out, int curve, const int* devices, int ndevices);
/* What the harness shims call: MI355_MSM_DEVICES = "0,1,2,3" | "0-7" | "all" selects a sharded context over those devices,
 * one entry (or unset) an ordinary one.  mi355_msm() (stateless) goes through here as well.
 * MI355_MSM_ASSUME_SUBGROUP = 0 | 1 sets the option "assume_subgroup" on the new context (the yrrid shim turns it on unless this says 0:
 * the reference it stands in for folds scalars with the top bit set, CMB ProcessSignedDigits.cu:123-128). */
RustError mi355_msm_create_env(mi355_msm_ctx** out, int curve);

/* Bases in HOST memory (arkworks Affine images, `stride` bytes apart).  Copies; caller keeps ownership. */
RustError mi355_msm_set_bases(mi355_msm_ctx* ctx, const void* affine, size_t npoints, size_t stride);
/* Same, bases already resident in DEVICE memory (e.g. a torch uint8 tensor's data_ptr). */
RustError mi355_msm_set_bases_device(mi355_msm_ctx* ctx, const void* d_affine, size_t npoints, size_t stride);

/* Bases as arkworks CanonicalSerialize UNCOMPRESSED records in host memory (row f2: what the harness persists with
 * `points.serialize_unchecked(File::create("points.bin"))`, P1B hardcaml/.../test_fpga_harness/src/util.rs:126-140): per point
 * x | y as little-endian normal-form integers (2 x 48 B for G1, 2 x 96 B for G2), SWFlags in the top two bits of the last
 * byte (bit 6 = infinity).  Pass the records WITHOUT the leading u64 element count.  Converted on the device. */
RustError mi355_msm_set_bases_serialized(mi355_msm_ctx* ctx, const void* records, size_t npoints);
/* The inverse for results: a Projective image (any Z) -> one uncompressed CanonicalSerialize record (host arithmetic),
 * comparable byte-for-byte with the harness's `arkworks_results.bin` entries. */
RustError mi355_msm_point_to_serialized(int curve, const void* projective, void* out_record);

/* `batches` MSMs over the SAME bases: scalars holds batches * npoints entries, out receives `batches`
 * projective images (P1A 6block/src/lib.rs:85-109: batch_size = scalars.len() / points.len()).
 * npoints may be smaller than the number of uploaded bases (prefix). */
RustError mi355_msm_run(mi355_msm_ctx* ctx, void* out_projective, const void* scalars, size_t npoints, size_t batches);
/* Scalars already in DEVICE memory; `stream` is the hipStream_t on which the scalars become ready and on which all work is
 * enqueued (NULL = the HIP default stream, i.e. ordered after whatever a framework's default stream produced).  `out_projective` is HOST memory; the call returns
 * after the result is written (one stream synchronisation per batch chunk). */
RustError mi355_msm_run_device(mi355_msm_ctx* ctx, void* out_projective, const void* d_scalars, size_t npoints,
                               size_t batches, void* stream);

/* ---- stream-ordered run -------------------------------------------------------------------------------------------------
 * Replaces the asynchronous half of the second-place entry's API: msm_configuration.stream, h2d_copy_finished{,_callback},
 * d2h_copy_finished{,_callback} and msm_execute_async (ML bellman-cuda.h:48-75, ML msm.cu:97-468; caller:
 * P1A matter-labs/src/lib.rs:150-190), which lets a prover keep its own kernels (NTTs) running while an MSM is in flight.
 *
 * mi355_msm_run_async returns as soon as the job is queued (no device synchronisation in the calling thread).  The MSM is ordered
 * AFTER everything enqueued in `stream` before the call (an event is recorded there; `d_scalars` must be valid from that point
 * until the job has finished) and runs on the context's own stream, so work the caller enqueues on ANY of its streams afterwards
 * overlaps it.  Jobs of one context run one after the other in submission order, on a worker thread the context owns -- the tail
 * of an MSM is host arithmetic (window fold, normalisation) and host decisions (out-of-memory back-off, the XYZZ repeat of an
 * Edwards run), which is why completion is a host-side event: `done(user, status)` is called from that thread once
 * `out_projective` (HOST memory, `batches` images) is written -- status.code 0 = success; status.message, if any, is the callback's
 * to free() (SPK util/rusterror.h:15-27) -- and / or the job handle can be polled and waited for.  At least one of `done` and `job`
 * must be given.  mi355_msm_job_wait blocks until the job has finished, returns its status and RELEASES the handle (call it
 * exactly once per handle); mi355_msm_job_done polls (1 = finished).  The synchronous entry points of the same context must not be
 * called while jobs are pending (query "async_pending"); mi355_msm_destroy runs pending jobs to completion first. */
typedef struct mi355_msm_job mi355_msm_job;
typedef void (*mi355_msm_done_fn)(void* user, RustError status);
RustError mi355_msm_run_async(mi355_msm_ctx* ctx, void* out_projective, const void* d_scalars, size_t npoints, size_t batches,
                              void* stream, mi355_msm_done_fn done, void* user, mi355_msm_job** job);
int mi355_msm_job_done(mi355_msm_job* job);
RustError mi355_msm_job_wait(mi355_msm_job* job);

/* "precompute" = 2 (auto, set BEFORE set_bases): the context picks the table levels itself from the device memory that is free at
 * set_bases -- a level per window, else 6, 4 or 3 levels (the shapes profiles/r04_table_levels_sweep.txt shows as wins: -6 % / -3 % /
 * -1.4 % / -1 % at 2^26 pairs), each only if it fits with its build temporaries and leaves the work buffers of a full chunk plus a
 * tenth of the device to the caller; between 2^18 pairs and 2^20 (BLS12-377 G1), 2^19 (BLS12-381 G1), 2^21 (G2) six levels, where
 * they are worth 6 - 23 % (profiles/r06_size_sweep_tables.txt); none at the sizes in between, below 2^18, or when memory is short
 * (about 64 GB free at 2^26), and a build that
 * fails all the same leaves the context on the table-free path.  mi355_msm_query "table_levels" says what it chose.  The harness
 * shims take it from the environment: MI355_MSM_PRECOMPUTE=auto|0|1 (the reference's init builds its tables untimed,
 * CMB MSM.cu:380-383).
 * "precompute" = 1 (set BEFORE set_bases) makes the context store the tables 2^(c w) * P_i for every window w, so all digits
 * of a scalar share one bucket set and the bucket->window reduction and the window fold shrink by the number of windows --
 * the fixed-base trick of the ZPrize winners (CMB PrecomputePoints.cu:10-39; P1A matter-labs/src/lib.rs:101-114), paid for in
 * the untimed init and in HBM (windows x 128 B per base: 94 GB at 2^26, 151 GB with the Edwards records).  Results are identical.
 * "table_levels" = k (with "precompute", BEFORE set_bases; default 0 = a level per window) builds only k levels 2^(c G j) * P_i,
 * j < k, for G = ceil(windows / k) bucket sets: window g + G j reads level j into bucket set g -- the reference's own shape is
 * k = 6 levels and 2 bucket sets for its 23-bit windows (CMB PrecomputePoints.cu:10-39, MSM.cu:380-383).  k times the base memory
 * instead of `windows` times; measured at 2^26 (profiles/r04_table_levels_sweep.txt): all levels 101 ms / 151 GB, k = 6 105 ms /
 * 86 GB, k = 3 107 ms / 47 GB, no tables 108 ms / 21.5 GB.
 * Tuning knobs ("window_bits" 2..24, "lane_entries", "max_chunk" <= 2^27, "seg_entries" >= 4, "reduce_log_chunk" /
 * "reduce_log_chunk0" 1..7: bucket-reduction chunk sizes on all / the first level); 0 restores the automatic choice.
 * ("lane_entries" = 0: 2^20 lanes up to 2^25 pairs, then ~512 entries per lane -- fitted so that the working blocks of the accumulate
 * launch fill their last generation of 256 blocks, one per CU: the launch takes ceil(blocks / 256) generations, profiles/r06_ab_lane_groups.txt.)
 * "reduce_scan" = 0 keeps the bucket reduction on the recursive chunked scheme only (default: its tail is a parallel scan);
 * "reduce_scan_log" 6..18 = log2 of the elements per window at which the scan takes over (default 12).
 * "quad_limit" (per context, default 2^18): merge / scan launches of at most that many additions spread each
 * addition over four lanes (latency); 0 = always one lane per addition.
 * "g2_paired" (G2 contexts; bit mask, default 31 = all; ignored on G1): which throughput kernels hold every Fp2 value on TWO
 * neighbouring lanes (c0 on the even, c1 on the odd one; csrc/fp2pair.hpp) and so run two waves per SIMD instead of one --
 * bit 0 bucket accumulation, 1 first level of the bucket reduction, 2 fragment merge, 3 scan steps, 4 bucket merge of carried
 * batches.  Same records in memory, same result bytes; 0 restores the one-lane-per-point kernels (the A/B of
 * profiles/r05_ab_g2_paired.txt: 2^24 pairs 122 -> 109 ms).
 * "reduce_fill" 1..4 (default 1): waves per SIMD the first chunked level of the bucket reduction is cut for (measured: 2 does
 * not pay, profiles/r05_ab_reduce_fill.txt).
 * "assume_subgroup" = 1 (default 0): the caller guarantees that every base lies in the order-r subgroup (r P = O), as the ZPrize
 * generator's do.  A scalar k in (r/2, r) then runs as (r - k)(-P): the winners' top-bit trick (CMB ProcessSignedDigits.cu:10-20,
 * 123-128), one significant bit less, so BLS12-377 scalars tile 12 windows of 21 bits and the auto window size moves from 20 to 21
 * at 2^26 pairs (-5 % additions; below 2^25 pairs the window choice is left alone).  Off by default because arkworks' msm is exact for ANY curve point and this is not.
 * "anchor" (default 1; 0 off; 2 = always, a test setting): the anchored window.  Signed digits carry, so where the window size leaves
 * (almost) no scalar bits above the last full window -- BLS12-377: 253 = 11 x 23, 252 = 12 x 21 -- the window above it is still
 * non-zero for 14-57 % of the scalars.  assume_subgroup removes those additions by an assumption; this removes them by arithmetic:
 * the carry chain ENDS at the last full window (its value v in [0, 2^c] is taken as 2^(c-1) + s, |s| <= 2^(c-1): the same buckets)
 * and the constant part, 2^(c a + c - 1) x (the plain sum of the bases of the run), is added on the host.  That sum is computed by
 * the library itself (k_sum_bases + the fragment merge: ~n mixed additions, 8 ms at 2^26) in set_bases for runs over all the bases, on
 * the first run of any other length, and kept until the next set_bases.  Exact for ANY input (tests/test_gpu_anchor.py: non-canonical scalars, points outside the subgroup).
 * Used for batches of >= 2^20 pairs, with "carry" on and "assume_subgroup" off, at the window sizes where it saves >= 1 % of the
 * additions; 2^26 pairs of BLS12-377 G1 then run at c = 21 instead of 20 (-1.0 %; with tables -1..-2.5 %: profiles/r06_ab_anchor.txt).
 * The price: a scalar of ZERO costs one addition (its digit in the anchored window is -2^(c-1)) instead of none -- a batch that is
 * mostly zeros should set "anchor" = 0.  The stateless call never uses it (its bases change with every call).
 * "carry" (default 1): a batch that runs as several chunks (max_chunk, the memory budget, the pieces of a host-scalar batch) carries
 * ONE bucket array through them -- every chunk uses the window size of the whole batch and only the last one reduces; 0 = every
 * chunk reduces its own buckets and the partial sums are added on the host.  "first_piece_div" (default 13; 4 with carry = 0): the
 * first batch of a host-scalar run is handed over as pieces of n/div, 3n/div, 9n/div ... each computed while the next crosses PCIe
 * (CMB MSM.cu:419-434 splits its first copy 1/4 + 3/4).
 * Test hooks: "mem_limit" (bytes of device memory chunks may be planned against), "inject_alloc_failures" (N > 0: the next N
 * work-buffer reservations fail; -K: only the K-th from now).
 * "scalars_montgomery" = 1 makes every run treat the scalars as arkworks `Fr` values (Montgomery form, a*2^256 mod r)
 * and convert them on the device first -- VariableBaseMSM::msm(bases, &[Fr]) = into_bigint + msm_bigint
 * (ARK ec/src/msm/variable_base/mod.rs:48-53; sppark's `mont` flag SPK msm/pippenger.cuh:157-164).
 * Mirrors Matter Labs' runtime msm_configuration (P1A matter-labs/.../bellman-cuda.h:49-71). */
RustError mi355_msm_set_option(mi355_msm_ctx* ctx, const char* key, long value);
/* "twisted_edwards" (default 1; set BEFORE set_bases; BLS12-377 G1 only) lets the context keep, next to the bases, their image on
 * the birationally equivalent twisted Edwards curve -X^2 + Y^2 = 1 + d X^2 Y^2 and accumulate there: 7 field multiplications
 * per mixed addition instead of 8M + 2S, no doubling/infinity branches (the trick of P1A Trapdoor-Tech/msm_opt.md and of the FPGA
 * entries).  Results are identical BY CONSTRUCTION, not by assumption: a base set containing one of the five points without
 * an image (e.g. the FPGA harness's 2-torsion fixture) stays on the short-Weierstrass path, and because d is a square the
 * kernels check every addition for a vanishing denominator (possible only for inputs outside the prime-order subgroup) and
 * the run is then repeated on the short-Weierstrass path.  Costs 192 B per base (per table level) of HBM on top. */
/* State of a context: "twisted_edwards" (1 = the current bases run on the twisted-Edwards path), "twisted_edwards_fallbacks"
 * (chunks repeated on the XYZZ path so far), "twisted_edwards_demotions" (two fallbacks in a row demote the base set to XYZZ
 * until the next set_bases), "oom_backoffs" (chunks restarted at half size after a device allocation failed -- after the idle
 * contexts of the stateless pool were given back and the same chunk retried), "debug_checks" (invariant checks a -DMSM_DEBUG
 * build has run; always 0 in this library), "chunk_cap", "anchor" (the option), "anchored_window" (1 + the anchored window of the most
 * recent chunk, 0 = plain digits), "anchor_sums" (sums of bases computed so far), "anchor_sum_us" (host time the most recent run spent on one),
 * "device", "bases", "table_levels", "table_window_bits", "base_bytes" (device bytes held for the bases). */
RustError mi355_msm_query(mi355_msm_ctx* ctx, const char* key, uint64_t* value);
/* Per-stage device time (ms, HIP events on the launch stream) of the most recent run, summed over its chunks
 * and batches (MI355_T_HOST_FOLD: host wall time of the final normalisation); ms must hold MI355_T_COUNT floats.
 * info (8 words): [0]=window bits, [1]=windows, [2]=sorted entries of the last chunk, [3]=entries per lane, [4]=accumulate
 * launches, [5]=lanes of the last launch, [6]=1 when precomputed tables were used, [7]=1 on the twisted-Edwards path.
 * Chunks are sized to the device memory that is free (the reference plans its allocations first, ML msm.cu:453-466), and a
 * chunk whose allocation fails all the same is retried at half the size. */
RustError mi355_msm_last_timings(mi355_msm_ctx* ctx, float* ms, uint64_t* info);
/* The same for ONE shard of a sharded context (mi355_msm_last_timings reports the slowest shard per stage): an imbalance
 * between the devices shows here.  shard 0 of an ordinary context is the context itself. */
RustError mi355_msm_shard_timings(mi355_msm_ctx* ctx, int shard, float* ms, uint64_t* info);

/* ---- stateless calls ---------------------------------------------------------------------------
 * sppark's mult_pippenger_inf(out, points, npoints, scalars, ffi_affine_sz)
 * (SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35) with the curve made explicit; BASELINE.json's
 * `msm(bases, scalars, n)` is mi355_msm(curve, out, bases, n, scalars, 104). */
RustError mi355_msm(int curve, void* out_projective, const void* affine, size_t npoints, const void* scalars,
                    size_t ffi_affine_sz);
/* The call is a pipeline, as in the reference (SPK msm/pippenger.cuh:617-661 uploads the next slice of points and scalars
 * while the current one is sorted and accumulated; CMB MSM.cu:419-434; the growing chunks of P1A matter-labs/src/lib.rs:171-182):
 * slices of 2^20..2^23 pairs are staged by a few host threads through a small ring of pinned buffers (kept for the life of the
 * process; mi355_msm_trim() gives it back) and cross PCIe while earlier slices are converted and run; partial sums are added
 * on the host.  Environment: MI355_MSM_STAGE_THREADS (default 6), MI355_MSM_STATELESS_SLICE_LOG (log2 pairs per slice),
 * MI355_MSM_STATELESS_RAMP (default 1: the first two slices are 1/8 and 1/2 of a slice; 0: one half slice first).
 * With MI355_MSM_DEVICES naming several GPUs every shard runs its own pipeline over its slice of both operands.
 * mi355_msm_last_stateless: what the calling thread's most recent stateless call did -- out[0..7] = total ms, setup ms (buffers,
 * ring, threads), ms the compute side waited for uploads, ms it spent issuing/awaiting slices, the last slice's share of that
 * (the tail nothing overlaps), slices, staging threads, bytes moved; out[8] = ms from the start of the call to the completion of its LAST
 * DMA (total ms - out[8] is what the call spends after the uploads are over: the bound is PCIe when that is one slice's compute,
 * the device when it is more), out[9] = ms at which the copy stream started. */
RustError mi355_msm_last_stateless(double* out, size_t count);
/* What the stateless entry points keep between calls, and its bounds: the pinned staging rings (12 x 16 MiB per device in use) and at
 * most ONE idle context per (curve, device) with its device buffers (~9 GB after a 2^26-pair G1 call; a context above
 * MI355_MSM_STATELESS_KEEP_MB, default 16384, is parked without its buffers).  Any device allocation of this library that runs out
 * of memory frees the idle contexts and retries before it fails.  mi355_msm_trim() frees rings and idle contexts now;
 * mi355_msm_pool_stats: out[0] = idle contexts, out[1] = device bytes they hold, out[2] = idle rings, out[3] = pinned bytes they hold. */
RustError mi355_msm_trim(void);
RustError mi355_msm_pool_stats(uint64_t* out, size_t count);

/* ---- arkworks' streaming accumulators (ARK ec/src/msm/variable_base/stream_pippenger.rs) ---------------------------------
 * ChunkedPippenger::{new, with_size, add, finalize} (:11-75) and HashMapPippenger::{new, add, finalize} (:78-140): what a prover
 * holds across rounds.  `hashmap` = 0: buffer pairs, and whenever the buffer holds max_msm_buffer of them, result += MSM(buffer).
 * `hashmap` = 1: a pair whose base (x, y, infinity flag) is already buffered adds its scalar to that entry modulo the scalar
 * field order r -- scalars are Fr values there -- and the MSM runs when max_msm_buffer DISTINCT bases are buffered.
 * add() takes `count` pairs (bases `stride` bytes apart) and behaves exactly like `count` single adds.  finalize() flushes what
 * is left, writes the normalised Projective image and leaves the accumulator empty for reuse (arkworks' finalize consumes it).
 * Each flush is the stateless pipeline of mi355_msm() on `device` (< 0: the current device).  Options: "scalars_montgomery"
 * (the scalars are arkworks Fr images, converted on the device -- sums of Montgomery images are Montgomery images of sums),
 * "window_bits".  Queries: "buffered", "flushes", "merged" (pairs that landed on an existing hashmap entry), "buf_size". */
typedef struct mi355_msm_stream mi355_msm_stream;
RustError mi355_msm_stream_create(mi355_msm_stream** out, int curve, int device, size_t max_msm_buffer, int hashmap);
RustError mi355_msm_stream_set_option(mi355_msm_stream* s, const char* key, long value);
RustError mi355_msm_stream_add(mi355_msm_stream* s, const void* affine, size_t stride, const void* scalars, size_t count);
RustError mi355_msm_stream_finalize(mi355_msm_stream* s, void* out_projective);
RustError mi355_msm_stream_query(mi355_msm_stream* s, const char* key, uint64_t* value);
RustError mi355_msm_stream_destroy(mi355_msm_stream* s);

/* Sum `count` projective images (any Z) into one normalised image: the multi-GPU combine step
 * ("final 8-point curve add").  Pure host arithmetic on <= a few dozen points; no device needed. */
RustError mi355_msm_fold(int curve, void* out_projective, const void* projective, size_t count);

/* Synthetic bases in the shape of the reference harness generator (P1A yrrid/src/util.rs:15-28): `distinct`
 * subgroup points (h0 + j*h1)*G derived from `seed`, written as arkworks Affine images `stride` bytes apart into
 * HOST memory and replicated by doubling the vector up to `npoints`.  Host arithmetic; no device needed. */
RustError mi355_msm_generate_points(int curve, uint64_t seed, size_t distinct, size_t npoints, void* out_affine,
                                    size_t stride);

/* The execution plan the engine would use for an MSM of `npoints` pairs (pure host arithmetic, no device): out[0..9] =
 * window bits, digit windows, windows owning buckets (1 with precompute), sorted entries, entries per lane, lanes,
 * fragment-merge launches, bucket-reduce launches, sort key bits, bytes of per-run device work buffers.
 * `options` may be NULL or {window_bits, lane_entries, seg_entries} (0 = automatic).  `precompute`: 0 = no tables, 1 = a table
 * level per window, k > 1 = the context option "table_levels" = k -- the plan then equals what a context with those options
 * reports through mi355_msm_query "table_window_bits" / "table_levels".
 * NOTE: this argument is a LEVEL COUNT, not the context option "precompute" (where 2 means "auto"): passing that option's value here
 * plans two table levels.  To plan what an auto context runs, query its "table_levels" after set_bases and pass that. */
RustError mi355_msm_plan(int curve, size_t npoints, int precompute, const long* options, uint64_t* out);

/* The slice [*lo, *hi) of range(npoints) that shard `shard` of `nshards` owns in a sharded context (and in dist.py's
 * one-process-per-GPU path): ceil(npoints / nshards) consecutive pairs per shard, the last ones possibly shorter or empty.
 * Pure host arithmetic. */
RustError mi355_msm_shard_bounds(size_t npoints, int nshards, int shard, size_t* lo, size_t* hi);

/* Library/ABI version and the gfx target the kernels were built for ("gfx950"). */
const char* mi355_msm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_MSM_H */
