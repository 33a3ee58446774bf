/*
 * vkmr_hip.h -- C ABI of the MI355X (gfx950) Merkle-root engine.
 *
 * This is the drop-in boundary: everything the reference's host code asked of
 * Vulkan for the map -> reduce -> combine hot path, as plain
 * C entry points over hand-written HIP kernels.  No C++ types, no exceptions, no
 * torch types.  Every function returns a vkmr_status (0 = ok, negative = error,
 * VKMR_NOT_READY = 1 from vkmr_hip_event_query only) and writes results through
 * out-pointers; vkmr_hip_last_error() gives the text of the last failure on the
 * calling thread.
 *
 * Each entry point cites the reference interface it replaces (file:line relative
 * to the reference tree).  INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Threading: one host thread may drive any number of devices; every call that
 * takes `dev` selects that device itself.  Calls on one instance are not
 * re-entrant across threads (same as the reference, which is single-threaded).
 *
 * Ownership: buffers passed to an *_async call stay owned by the caller and must
 * stay alive until an event recorded after the call on the same stream has
 * completed (reference: a Batch is freed when its Mapping retires,
 * src/vkmr/Mappings.cpp:328-329; a Slice when its Reduction retires,
 * src/vkmr/Reductions.cpp:663).
 */
#ifndef VKMR_HIP_H
#define VKMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKMR_API __attribute__((visibility("default")))

/* ---- status ------------------------------------------------------------- */
typedef int vkmr_status;
#define VKMR_OK            0
#define VKMR_NOT_READY     1   /* vkmr_hip_event_query: work still in flight (VK_NOT_READY) */
#define VKMR_ERR_INVALID  (-1) /* bad argument (null pointer, count/height mismatch, ...)   */
#define VKMR_ERR_NO_DEVICE (-2)
#define VKMR_ERR_OOM      (-3) /* VK_ERROR_OUT_OF_DEVICE_MEMORY / _HOST_MEMORY              */
#define VKMR_ERR_HIP      (-4) /* any other HIP runtime failure                              */
#define VKMR_ERR_COMM     (-5) /* RCCL failure, or librccl could not be loaded               */

/* ---- wire structs: identical layout to the reference's device structs ---- */

/* One packed input string.  Replaces VkSha256Metadata, src/common/SHA-256defs.h:51-54:
 * `start` is a 32-bit WORD index into the packed data buffer, `size` is in BYTES. */
typedef struct vkmr_metadata { uint32_t start; uint32_t size; } vkmr_metadata;

/* One digest cell.  Replaces VkSha256Result, src/common/SHA-256defs.h:47-49: eight
 * words holding the VALUES H[0..7], so raw little-endian memory is byte-swapped per
 * word relative to the canonical digest. */
typedef struct vkmr_digest { uint32_t data[8]; } vkmr_digest;

typedef struct vkmr_stream_s* vkmr_stream;   /* replaces VkQueue + VkCommandBuffer      */
typedef struct vkmr_event_s*  vkmr_event;    /* replaces VkFence and the timestamp pair */

/* ---- devices (replaces VkSha256D's enumeration, src/vkmr/SHA-256vk.cpp:38-171) ---- */
VKMR_API vkmr_status vkmr_hip_device_count(int* count);
/* Marketing name of device `dev` (VkPhysicalDeviceProperties::deviceName). */
VKMR_API vkmr_status vkmr_hip_device_name(int dev, char* buf, size_t buflen);
/* Free/total HBM in bytes (ComputeDevice::AvailableMemoryTypes budgets,
 * src/vkmr/Devices.h:184-245; used for slice sizing, src/vkmr/Slices.h:421-454). */
VKMR_API vkmr_status vkmr_hip_device_mem_info(int dev, size_t* free_bytes, size_t* total_bytes);
/* Compute-unit count and wavefront width (64 on gfx950); the reference reads
 * subgroupSize here, src/vkmr/Reductions.cpp:749-770. */
VKMR_API vkmr_status vkmr_hip_device_geometry(int dev, int* compute_units, int* wavefront);

/* ---- memory --------------------------------------------------------------- */
/* Pinned, zero-filled host buffer for a Batch's data / metadata (Batch::Buffer,
 * src/vkmr/Batches.cpp:196-237: host-visible + coherent, memset 0). */
VKMR_API vkmr_status vkmr_hip_host_alloc(size_t bytes, void** out);
VKMR_API vkmr_status vkmr_hip_host_free(void* p);
/* Device-local buffer: a Slice of digests (Slices::New, src/vkmr/Slices.h:297-384)
 * or the HBM landing zone of a batch. */
VKMR_API vkmr_status vkmr_hip_device_alloc(int dev, size_t bytes, void** out);
VKMR_API vkmr_status vkmr_hip_device_free(int dev, void* p);
VKMR_API vkmr_status vkmr_hip_memset_async(int dev, vkmr_stream s, void* dst, int value, size_t bytes);
VKMR_API vkmr_status vkmr_hip_memcpy_h2d_async(int dev, vkmr_stream s, void* dst_dev, const void* src_host, size_t bytes);
/* Used for the 32-byte root read-back (vkCmdCopyBuffer slice[0] -> host buffer,
 * src/vkmr/Reductions.cpp:537-540) and by tests to fetch digests. */
VKMR_API vkmr_status vkmr_hip_memcpy_d2h_async(int dev, vkmr_stream s, void* dst_host, const void* src_dev, size_t bytes);

/* ---- streams and events ---------------------------------------------------- */
/* A stream orders the ops submitted to it (the compute->compute barriers of
 * src/vkmr/Reductions.cpp:506-519 come for free).  ComputeDevice::Queue round-robin,
 * src/vkmr/Devices.cpp:525-538. */
VKMR_API vkmr_status vkmr_hip_stream_create(int dev, vkmr_stream* out);
VKMR_API vkmr_status vkmr_hip_stream_destroy(int dev, vkmr_stream s);
/* Start-up work, done when the caller wants it rather than inside its first copy and first launch: the library's
 * kernels are loaded onto the device and the stream's queue comes up (VKMR_WARM_KERNELS: one launch of the map kernel on
 * a four-byte string), the copy engine comes up (VKMR_WARM_COPY: one host-to-device copy of `copy_bytes` from pinned
 * memory on the stream -- the engine's set-up depends on the size: give the size of the copies to come, at least 256);
 * returns when both are done.  The reference pays the same at start-up, outside its stopwatch: ComputeDevice builds its
 * shader modules and compute pipelines before run() begins (src/vkmr/Devices.cpp:225-280, Shaders.cpp:20-40).
 * Optional: without it the first vkmr_hip_memcpy_h2d_async and the first launch take the time (20 + 15 ms on MI355X,
 * profiles/r03_frontend_phases.txt). */
#define VKMR_WARM_KERNELS 1u
#define VKMR_WARM_COPY 2u
VKMR_API vkmr_status vkmr_hip_warm_up(int dev, vkmr_stream s, unsigned what, size_t copy_bytes);
VKMR_API vkmr_status vkmr_hip_stream_sync(int dev, vkmr_stream s);
VKMR_API vkmr_status vkmr_hip_event_create(int dev, vkmr_event* out);
VKMR_API vkmr_status vkmr_hip_event_destroy(int dev, vkmr_event e);
VKMR_API vkmr_status vkmr_hip_event_record(int dev, vkmr_event e, vkmr_stream s);
/* vkGetFenceStatus (src/vkmr/Mappings.cpp:322, Reductions.cpp:642): VKMR_OK or VKMR_NOT_READY. */
VKMR_API vkmr_status vkmr_hip_event_query(int dev, vkmr_event e);
/* vkWaitForFences (src/vkmr/Mappings.cpp:362, Reductions.cpp:686). */
VKMR_API vkmr_status vkmr_hip_event_wait(int dev, vkmr_event e);
/* Make stream `s` wait for `e` (cross-stream ordering; no Vulkan counterpart needed
 * in the single-queue reference). */
VKMR_API vkmr_status vkmr_hip_stream_wait_event(int dev, vkmr_stream s, vkmr_event e);
/* QueryPoolTimer::ElapsedMillis, src/vkmr/QueryPoolTimers.cpp:52-93. */
VKMR_API vkmr_status vkmr_hip_event_elapsed_ms(int dev, vkmr_event begin, vkmr_event end, float* ms);

/* ---- the hot path ---------------------------------------------------------- */

/*
 * MAP: digests[i] = SHA-256(SHA-256(string i)) for i in [0,count).
 * Replaces Mapping::Dispatch + shader entry `_SHA_256_N_`
 * (src/vkmr/Mappings.cpp:135-232, src/shaders/SHA-256.comp:177-304).
 *   data_dev    packed words in HBM, layout of Batch::Push (src/vkmr/Batches.cpp:64-121)
 *   data_words  number of valid 32-bit words in data_dev (bounds every load)
 *   meta_dev    count entries; string i occupies bytes [4*start, 4*start+size)
 *   out_dev     count digest cells (a sub-slice, src/vkmr/Slices.h:145-187)
 * Deliberate differences from the shader: bounds test is `>=`
 * (Q4), tail bytes of the last word are masked to `size` (Q3), the 64-bit length
 * uses size>>29 for the high word.  size == 0 hashes the empty string.  A string whose
 * metadata runs past data_words is cut at the end of the buffer (bounded work for
 * corrupt metadata; the shader would read out of bounds).
 */
VKMR_API vkmr_status vkmr_hip_map_async(int dev, vkmr_stream s,
                                        const uint32_t* data_dev, uint64_t data_words,
                                        const vkmr_metadata* meta_dev, uint32_t count,
                                        vkmr_digest* out_dev);

/*
 * REDUCE: sub-tree root of `count` digests through exactly `height` levels of
 * node = SHA-256d(left || right), an unpaired node being paired with itself at
 * every level, including after the count has collapsed to one.
 * Replaces Reduction::Apply + ReductionBySubgroup::GetCommandBuffer + shader entry
 * `_SHA_256_2_BE_`/`_VKMR_BY_SUBGROUP_` (src/vkmr/Reductions.cpp:147-216, :433-547,
 * src/shaders/SHA-256.comp:308-391).  The caller chooses `height` exactly as the
 * reference chooses `applicable` (src/vkmr/Reductions.cpp:471): log2(capacity) for
 * every slice when the stream spans several slices, ceil(log2(count)) (at least 1:
 * a lone leaf is hashed with itself, the CPU backend's rule)
 * for a single slice.  Requires ceil(count / 2^height) == 1.
 *   digests_dev  count cells, read only (the reference reduces in place)
 *   scratch_dev  vkmr_hip_reduce_scratch_bytes(count) bytes of device memory; the size function is an
 *                upper bound for EVERY run of at most `count` digests, so scratch sized for a slice's
 *                capacity serves any shorter slice (may be NULL for count <= 128)
 *   root_dev     one cell in device memory receiving the root
 * height is at most 63.
 */
VKMR_API vkmr_status vkmr_hip_reduce_async(int dev, vkmr_stream s,
                                           const vkmr_digest* digests_dev, uint64_t count, uint32_t height,
                                           void* scratch_dev, vkmr_digest* root_dev);
VKMR_API size_t vkmr_hip_reduce_scratch_bytes(uint64_t count);

/*
 * METADATA FROM SIZES.  The strings of a batch lie back to back (string i + 1 starts on the word after string i:
 * Batch::Push, src/vkmr/Batches.cpp:64-121), so entry i is {first_word + sum over j < i of ceil(size[j] / 4), size[i]}:
 * what the reference's host code computes while it appends (WordCount, Batches.cpp:182-187).  A caller that feeds the
 * device over PCIe may send the 16-bit sizes (2 bytes per string instead of 8) and have the entries written in device
 * memory, where vkmr_hip_map_async reads them.  Every size must be below 65 536 (send the entries themselves otherwise).
 *   sizes_dev    count sizes, 16-byte aligned
 *   scratch_dev  vkmr_hip_sizes_scratch_bytes(count) bytes
 *   meta_dev     count entries, 16-byte aligned, written
 */
VKMR_API vkmr_status vkmr_hip_metadata_from_sizes_async(int dev, vkmr_stream s, const uint16_t* sizes_dev, uint32_t count,
                                                        uint32_t first_word, void* scratch_dev, vkmr_metadata* meta_dev);
VKMR_API size_t vkmr_hip_sizes_scratch_bytes(uint32_t count);

/*
 * REDUCE, several slices at once: `nslices` consecutive slices of `capacity` digests
 * each (a power of two), the last holding `count_last` <= capacity, all reduced
 * through `height` levels by the same launches; roots_dev[k] receives slice k's
 * root.  This is Instance::Root's loop "for every remaining slice Reduce(...)"
 * (src/vkmr/SHA-256vk.cpp:301-311) issued as one operation, so that the
 * latency-bound tops of the sub-trees run side by side instead of one after the
 * other.  Same per-slice contract as vkmr_hip_reduce_async.  scratch_dev needs
 * vkmr_hip_reduce_slices_scratch_bytes(capacity, nslices).
 */
VKMR_API vkmr_status vkmr_hip_reduce_slices_async(int dev, vkmr_stream s,
                                                  const vkmr_digest* digests_dev, uint32_t nslices,
                                                  uint64_t capacity, uint64_t count_last, uint32_t height,
                                                  void* scratch_dev, vkmr_digest* roots_dev);
VKMR_API size_t vkmr_hip_reduce_slices_scratch_bytes(uint64_t capacity, uint32_t nslices);

/*
 * PROOF (the reference's own "to do", README.md:118-120): the authentication path of
 * leaf `index` in exactly the tree vkmr_hip_reduce_async(count, height) computes.
 * siblings_dev[l], l = 0..height-1, receives the node the path node is hashed with at
 * level l: its left or right neighbour (bit l of `index` says which side the path node
 * is on: 0 = path node left), or the path node itself where it has no right sibling
 * (duplicate-last rule).  Folding the leaf digest with the siblings from l = 0 upwards
 * gives the root, which is also written to root_dev (may be NULL).  For a stream that
 * spans several slices the caller chains two proofs: this one inside the slice
 * (height = log2 capacity) and one over the slice roots.  Each sibling is the root of
 * the neighbouring sub-tree of 2^l leaves and is computed with the same kernels as the
 * reduction (total work: about one more reduction of the slice).
 * scratch_dev: vkmr_hip_reduce_scratch_bytes(count) bytes.
 */
VKMR_API vkmr_status vkmr_hip_proof_async(int dev, vkmr_stream s,
                                          const vkmr_digest* digests_dev, uint64_t count, uint32_t height,
                                          uint64_t index, void* scratch_dev,
                                          vkmr_digest* siblings_dev, vkmr_digest* root_dev);

/*
 * REDUCE + PROOFS in one pass -- the reference's to-do as it words it: "for an indicated leaf,
 * allocate a buffer to hold and write out the intermediate values during reduction"
 * (README.md:118-120).  Exactly vkmr_hip_reduce_async (same launches, same root) with the
 * kernels' proof-writing instantiations: the lane that hashes the pair a path node belongs to
 * stores the other node of that pair.  indices[0..k): leaf indices (< count), k <= 16, read on
 * the host at call time.  siblings_dev[q * height + l] = sibling of indices[q]'s path node at
 * level l, as vkmr_hip_proof_async defines it; no extra hash is computed, the reduction's time
 * does not change measurably.  vkmr_hip_proof_async is kept as the independent cross-check.
 */
VKMR_API vkmr_status vkmr_hip_reduce_proofs_async(int dev, vkmr_stream s,
                                                  const vkmr_digest* digests_dev, uint64_t count, uint32_t height,
                                                  void* scratch_dev, vkmr_digest* root_dev,
                                                  const uint64_t* indices, uint32_t k, vkmr_digest* siblings_dev);

/*
 * One tree level per launch, one lane per pair: the reference's non-subgroup
 * reduction (BasicReduction, src/vkmr/Reductions.cpp:257-409; shader :393-434).
 * Kept as an independent cross-check of vkmr_hip_reduce_async; same contract.
 * scratch_dev needs vkmr_hip_reduce_levels_scratch_bytes(count).
 */
VKMR_API vkmr_status vkmr_hip_reduce_levels_async(int dev, vkmr_stream s,
                                                  const vkmr_digest* digests_dev, uint64_t count, uint32_t height,
                                                  void* scratch_dev, vkmr_digest* root_dev);
VKMR_API size_t vkmr_hip_reduce_levels_scratch_bytes(uint64_t count);

/*
 * COMBINE: duplicate-last Merkle root over n >= 1 slice roots given in slice order, always
 * at least one level -- the rule of CpuSha256D::Root that the reference applies to the slice
 * roots on the CPU (CpuSha256DforReductions, src/vkmr/Reductions.cpp:56-69, :703-712).  Here
 * the roots stay in HBM (where the reductions or the gather below left them) and the combine is
 * one more launch on the caller's stream: no allocation, no synchronisation.  For n == 1 the
 * reference prints the slice root itself (src/vkmr/Reductions.cpp:692-701); callers do the same
 * and do not call combine.
 *   scratch_dev  vkmr_hip_reduce_scratch_bytes(n) bytes; may be NULL for n <= 128
 */
VKMR_API vkmr_status vkmr_hip_combine_async(int dev, vkmr_stream s, const vkmr_digest* roots_dev, uint32_t n,
                                            void* scratch_dev, vkmr_digest* root_dev);

/* ---- multi-GPU: slices sharded over devices, ONE gather of the sub-tree roots --------------
 *
 * The reference drives one device and collects slice roots on the host: each Reduction copies
 * its 32-byte root into a host-visible buffer (src/vkmr/Reductions.cpp:125-145, :537-540) and
 * ReductionsImpl::WaitFor feeds them, in slice order, to CpuSha256DforReductions (:56-69,
 * :703-712).  With slices sharded over the GPUs of a node that collection is a single RCCL
 * all-gather over xGMI (32 bytes per slice), after which any rank holds every root and runs
 * vkmr_hip_combine_async.  Two ways to form the communicator: one process driving every GPU
 * (the C++ front end's "hip:all"), or one process per GPU (bench.py under torch.distributed.run;
 * the 128-byte id travels over whatever channel the host program already has).
 * librccl is loaded on the first communicator, never by single-GPU runs.
 */
typedef struct vkmr_comm_s* vkmr_comm;
#define VKMR_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */

/* One process, ndev devices: rank i of the communicator is devs[i]. */
VKMR_API vkmr_status vkmr_hip_comm_init_all(const int* devs, int ndev, vkmr_comm* out);
/* One process per device: rank 0 creates the id, every rank joins with the same id. */
VKMR_API vkmr_status vkmr_hip_comm_create_id(void* id /* VKMR_COMM_ID_BYTES */);
VKMR_API vkmr_status vkmr_hip_comm_init_rank(int dev, const void* id, int nranks, int rank, vkmr_comm* out);
VKMR_API vkmr_status vkmr_hip_comm_destroy(vkmr_comm c);
/* Ranks in the communicator and how many of them this process drives. */
VKMR_API vkmr_status vkmr_hip_comm_size(vkmr_comm c, int* nranks, int* nlocal);

/*
 * GATHER: every rank contributes `per_rank` roots; when the call has completed on streams[i],
 * all_dev[i] (nranks * per_rank cells) holds rank r's roots at [r * per_rank, (r + 1) * per_rank).
 * The three arrays have one entry per LOCAL rank (one entry in a one-process-per-GPU program).
 * Ranks with fewer roots pad to per_rank; the padding is never read by the combine.
 * Issued as one ncclAllGather per local rank (grouped), enqueued on the given streams: ordered
 * after the reductions that produce roots_dev[i] when those ran on the same stream.
 */
VKMR_API vkmr_status vkmr_hip_gather_roots_async(vkmr_comm c, const vkmr_stream* streams,
                                                 const vkmr_digest* const* roots_dev, uint32_t per_rank,
                                                 vkmr_digest* const* all_dev);

/*
 * Reorders gathered roots into slice order when slices were dealt round-robin (slice k on rank
 * (k-1) % nranks, Slices::New -- the reference numbers slices 1, 2, ... in stream order,
 * src/vkmr/Slices.h:371): out[k] = gathered[(k % nranks) * per_rank + k / nranks], k < total.
 * With one slice per rank (bench.py) the gathered order already is the slice order.
 */
VKMR_API vkmr_status vkmr_hip_roots_in_slice_order_async(int dev, vkmr_stream s, const vkmr_digest* gathered_dev,
                                                         uint32_t nranks, uint32_t per_rank, uint32_t total,
                                                         vkmr_digest* out_dev);

/* Canonical lower-case hex of a digest cell (hash_to_string + print_bytes,
 * src/vkmr/SHA-256plus.cpp:453-469, src/vkmr/Debug.cpp:38-46).  hex holds 65 bytes. */
VKMR_API void vkmr_hip_digest_hex(const vkmr_digest* d, char* hex);

/* ---- diagnostics ----------------------------------------------------------- */
VKMR_API const char* vkmr_hip_last_error(void);
/* What the LAST vkmr_hip_map_async of this process launched (kernel instantiation, fetch mode, tile), the reduction
 * kernels, and " build=<id>": the identity of the kernel sources and build parameters this library was made from
 * (vk_merkle_roots_amd/build.py: source_id) -- profile records are matched against it (profiles/pmc_latest.json). */
VKMR_API const char* vkmr_hip_kernel_info(void);
/* Which RCCL the communicators are bound to: the shared object ncclAllGather lives in and ncclGetVersion, e.g.
 * "rccl=/opt/rocm/lib/librccl.so.1 version=22703"; "rccl=not loaded" before the first communicator.  A process that
 * carries PyTorch must show torch's copy here, not a second one (csrc/comm_rccl.hpp). */
VKMR_API const char* vkmr_hip_comm_info(void);

#ifdef __cplusplus
}
#endif
#endif /* VKMR_HIP_H */
