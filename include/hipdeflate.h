/*
 * hipdeflate.h -- C ABI of libhipdeflate.so: the MI355X (gfx950) block-parallel
 * DEFLATE engine that sits behind 7bgzf's codec boundary as BGZF_METHOD=hip.
 *
 * Plain C: pointers and sizes only.  Every entry point names the reference
 * interface it replaces or extends (paths relative to cielavenir/7bgzf):
 *
 *   hip_deflate / hip_inflate      = one more backend pair with the
 *       zlibutil_code_enc / zlibutil_code_dec signatures, lib/zlibutil.h:46-47,
 *       next to libdeflate_deflate / libdeflate_inflate (lib/zlibutil.h:101-114,
 *       lib/zlibutil.c:179-204).  DEFLATE_HIP extends the enum at
 *       lib/zlibutil.h:13-26.
 *   bgzf_compress                  = the LD_PRELOAD hook, bgzf_compress.c:39,
 *       with BGZF_METHOD=hip<level> added to its method table (:53-113).
 *   hipdeflate_batch_*             = the batch-shaped form of the per-block
 *       loop of applet/7bgzf.c:159-277 / applet/7migz.c:130-244 (encode) and
 *       applet/7bgzf.c:306-360 (decode): thousands of independent blocks per
 *       call instead of one pthread per block.  No reference counterpart
 *       exists; INTEGRATION.md shows the loop rewritten on top of them.
 *
 * Return convention everywhere: 0 = success, non-zero = failure, as the
 * reference's codecs (applet/7bgzf.c:228-254,350-353 only print the value).
 * There is NO CPU fallback: without a usable gfx950 device every entry point
 * fails (HD_E_NODEVICE) and says so on stderr.
 */
#ifndef HIPDEFLATE_H
#define HIPDEFLATE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* value to add after DEFLATE_KZIP in lib/zlibutil.h:13-26 */
#define DEFLATE_HIP 11

/* library error codes (positive; never collide with the 0..3 inflate codes) */
#define HD_E_NODEVICE  100   /* no HIP device / not gfx950 / runtime error     */
#define HD_E_ARG       101   /* bad argument (alignment, sizes)                 */
#define HD_E_NOMEM     102   /* device or pinned allocation failed              */

/* container framing the encode kernel writes around each payload */
#define HD_FRAME_RAW   0     /* raw DEFLATE only (what a zlibutil codec returns) */
#define HD_FRAME_BGZF  1     /* applet/7bgzf.c:263-272: 18 B header, CRC32, ISIZE */
#define HD_FRAME_MIGZ  2     /* applet/7migz.c:224-233: 20 B header, CRC32, ISIZE */
#define HD_FRAME_RAW_FLUSH 3 /* raw DEFLATE in full-flush form: no block is final, then an empty stored
                              * block header, byte alignment and 00 00 ff ff -- byte for byte what
                              * zlibutil_buffer_full_flush (applet/7dictzip.c:93-126, 7razf.c:126-160)
                              * makes of a codec's output by re-inflating it with a patched zlib; here it
                              * comes straight from the kernel.  Chunks in this form concatenate. */
#define HD_FRAME_ZLIB  4     /* RFC 1950 as zlibutil_buffer_code writes it (lib/zlibutil.c:374-397):
                              * 78 da, raw DEFLATE, Adler-32 big-endian -- the Adler-32 comes from the kernel */
#define HD_FRAME_GZIP  5     /* RFC 1952 as lib/zlibutil.c:379-405: 1f 8b 08 00 <mtime = 0> 02 00, raw
                              * DEFLATE, CRC32, ISIZE (the reference stamps time(NULL); a batch has no clock) */

/* OR'ed into `frame`: LATENCY MODE for batches far smaller than the machine -- what bgzf_compress, hip_deflate and
 * hip_deflate_flush use (one block per call, bgzf_compress.c:163-169, lib/zlibutil.c:179-192).
 *   levels 1..2  every block longer than 4080 (level 1) / 8160 (level 2) bytes is coded as independent flushed segments of
 *                that size, one wavefront each (level 2: four parse wavefronts per segment, HD_LAT_PART_BYTES), stitched on
 *                the device (hipdeflate_params.h HD_LAT_SEG_BYTES): a 0xff00-byte block in about a tenth of one wavefront's
 *                time.  The bytes differ from the throughput form (both are what the CPU twin gives for the same mode).
 *   levels 3..9  ONE CODEC PER LEVEL (round 5; as the reference has, deflate_compress.c:3951-3955): the flag changes the
 *                schedule, not the stream -- the workgroup parse on the whole block and, for blocks up to 64 KiB, the member
 *                written by a workgroup of sixteen wavefronts (hd_emit_wg.hpp) instead of by one; byte for byte the
 *                throughput form's output. */
#define HD_FRAME_LATENCY 0x100

/* ---- lifetime ---------------------------------------------------------- */

/* The device list (SURVEY.md 8(b) `hipdeflate_init(devices...)`; the reference's analogue is -@ N worker threads,
 * applet/7bgzf.c:155-217).  The library keeps one context -- stream, tables, scratch -- per ENTRY of the list; an entry
 * is a HIP device ordinal and an ordinal may be listed twice (two independent contexts on one card: how the
 * multi-device hosts are rehearsed on a one-GPU box).  The list is fixed by the first of:
 *   hipdeflate_init_devices(list, n)            an explicit list (n <= 32);
 *   hipdeflate_init(device)                     device >= 0: a list of one;
 *   hipdeflate_init(-1), or any other entry     HIPDEFLATE_DEVICES=0,1,2,... if set (a list), else a list of one:
 *   point, lazily                               HIPDEFLATE_DEVICE, else LOCAL_RANK (a torch.distributed / RCCL rank owns
 *                                               one card), else 0.
 * Contexts are created on first use.  A thread's calls run on entry 0 unless it chose another with
 * hipdeflate_use_device(index) (thread-local, like hipSetDevice); pipes and latency contexts stay on the entry they
 * were opened on whatever thread calls them; the LD_PRELOAD hook and the per-block codecs spread their batch contexts
 * over the whole list.  Idempotent; hipdeflate_init_devices with a list other than the one in force is HD_E_ARG. */
int  hipdeflate_init(int device);
int  hipdeflate_init_devices(const int *devices, int n);
int  hipdeflate_device_count(void);          /* entries of the list (configures it from the environment if need be) */
int  hipdeflate_use_device(int index);       /* 0, or HD_E_ARG / HD_E_NODEVICE; index = position in the list */
void hipdeflate_shutdown(void);              /* every context; the next call configures the list afresh */
/* 0 if a usable device is present and the kernels loaded, else HD_E_NODEVICE */
int  hipdeflate_available(void);
/* human-readable build/device description, never NULL */
const char *hipdeflate_version(void);
/* Workgroups of the parse kernel (levels >= 3) that have given a block up since the contexts were made, over all contexts: a
 * table turn that did not come within ~40 ms of polling (a preempted or single-stepped device) -- such a block is written
 * STORED: valid, status 0, but not the bytes an undisturbed run writes.  Also counted: an emit wavefront that waited 2 s for a block's
 * parse where the emit kernel runs beside the parse (launches of 512 and more blocks of up to 2 MiB at levels >= 3: the two kernels
 * need to run at the same time, on a stream of the lowest priority class the library makes for it; HIPDEFLATE_NO_BESIDE=1 turns the
 * scheme off) -- the same consequence.  0 in every healthy run; bench.py, the GPU tests and the fuzz tools assert it.
 * (Synchronises the devices.) */
uint64_t hipdeflate_stall_count(void);

/* ---- per-block codecs: drop-in zlibutil backends ------------------------ */

/* zlibutil_code_enc (lib/zlibutil.h:47).  *destLen in = capacity, out = bytes.
 * Output is raw DEFLATE ending in a BFINAL block.  Levels (include/hipdeflate_params.h): 0 stored; 1 greedy + static Huffman
 * (the speed level); 2 greedy + dynamic Huffman in one wavefront's 4 KiB window; 3..9 the workgroup parse -- a 32 KiB window and a
 * 64 KiB multi-way table shared by a workgroup, block splitting -- with 1 way greedy (3), 1 way lazy (4), 2 ways (5), 4 ways (6..9):
 * level 3 is below the reference's libdeflate level 1 in size on every measured set, level 6 within 3 % of its level 6 -- through
 * every entry point: the per-call forms here and the hook write the batch calls' bytes at levels >= 3 (HD_FRAME_LATENCY above).
 * As libdeflate_deflate (lib/zlibutil.c:179-192) a call succeeds whenever the stream fits the room, also for a block longer than the
 * room (applet/7png.c:112 gives 1.5 x the OLD compressed size).  One exception: hipdeflate_batch_deflate_dev at levels >= 3, where only
 * the device knows the lengths and the parse's records are sized by the slot, refuses (status != 0) a block longer than its slot.
 * Re-entrant and thread-safe; concurrent callers whose room covers the latency form's worst case and the stored form share launches
 * (the hook's micro-batcher, an engine per level and frame; HIPDEFLATE_CODEC_BATCH=0: a context per call). */
int hip_deflate(unsigned char *dest, size_t *destLen,
		const unsigned char *source, size_t sourceLen, int level);

/* zlibutil_code_dec (lib/zlibutil.h:46).  Stops at BFINAL, ignores trailing
 * source bytes (applet/7bgzf.c:328 passes payload + 8-byte trailer).  Returns
 * enum libdeflate_result values 0/1/3 like libdeflate_inflate
 * (lib/zlibutil.c:194-204).  Re-entrant and thread-safe, and built for the way the reference calls it -- a thread
 * per block, -@ N at once (applet/7bgzf.c:330-345): concurrent calls are coalesced into one launch (one wavefront
 * per stream, the whole window in LDS) on pinned batch memory, no process-wide lock; the batches are spread over the
 * device list.  At most HIPDEFLATE_INFLATE_INFLIGHT (2) batches are on the device, the collecting one grows meanwhile;
 * HIPDEFLATE_INFLATE_WINDOW_US (400) / _LINGER_US (60) bound how long a batch's first caller waits for the others
 * once a launch slot is free (a lone caller never waits). */
int hip_inflate(unsigned char *dest, size_t *destLen,
		const unsigned char *source, size_t sourceLen);

/* hip_deflate followed by zlibutil_buffer_full_flush (applet/7dictzip.c:93-126):
 * same contract, output in HD_FRAME_RAW_FLUSH form. */
int hip_deflate_flush(unsigned char *dest, size_t *destLen,
		      const unsigned char *source, size_t sourceLen, int level);

/* The decoder for such chunks -- the role zlib_inflate / igzip_inflate play in the
 * readers of 7dictzip (applet/7dictzip.c:318-323) and 7razf: a chunk has no final
 * block, so the stream may also stop after a non-final block once every source byte
 * has been used (lib/zlibutil.c:289-291, lib/zlibutil_igzip.c:111: "out of input" is
 * success there).  Input that runs out inside a block is still HD_BAD_DATA. */
int hip_inflate_flush(unsigned char *dest, size_t *destLen,
		      const unsigned char *source, size_t sourceLen);

/* Bytes of output room that always suffice for one block of block_bytes at `level`, in any frame,
 * a multiple of 16: the slot size (out_stride / out_cap) to give the batch calls, the role of the
 * 1.5 x block the reference allocates (zlibutil_buffer_allocate, applet/7bgzf.c:168).  Levels >= 1
 * code a block longer than HD_SEG_LIMIT in flushed 64 KiB segments (hipdeflate_params.h) and refuse
 * it when the room is below this bound, whatever the data would have needed. */
uint64_t hipdeflate_bound(uint64_t block_bytes, int level);

/* ---- batch API, host buffers ------------------------------------------- */

/* Compress nblocks independent blocks.  Block i is in[in_off[i] .. +in_len[i]).
 * Its output (framed as `frame` says) is written to out + i*out_stride, at most
 * min(out_stride, out_cap) bytes; out_len[i] = bytes written, crc32[i] = CRC-32
 * of the block's INPUT (fcrc32, applet/7bgzf.c:269), status[i] = 0 or 1 (does
 * not fit).  crc32/status may be NULL.  out_stride must be a multiple of 16.
 * Returns 0 if the batch ran (look at status[] per block), else HD_E_*. */
int hipdeflate_batch_deflate(const uint8_t *in, const uint64_t *in_off,
			     const uint32_t *in_len, uint32_t nblocks,
			     int level, int frame,
			     uint8_t *out, uint64_t out_stride, uint32_t out_cap,
			     uint32_t *out_len, uint32_t *crc32, int32_t *status);

/* Decompress nblocks independent raw-DEFLATE streams.  Stream i is
 * in[in_off[i] .. +in_len[i]) (trailing bytes allowed), its output goes to
 * out + out_off[i], capacity out_cap[i]; out_len[i] = bytes produced,
 * crc32[i] = CRC-32 of the OUTPUT (may be NULL), status[i] = 0/1/3 as
 * hip_inflate.  One stream must be shorter than HD_INFLATE_MAX_IN (2^28 bytes, hipdeflate_params.h):
 * the host entry points return HD_E_ARG for a longer one, the _dev entry points (which cannot see
 * in_len) report status 1 / out_len 0 for it. */
int hipdeflate_batch_inflate(const uint8_t *in, const uint64_t *in_off,
			     const uint32_t *in_len, uint32_t nblocks,
			     uint8_t *out, const uint64_t *out_off,
			     const uint32_t *out_cap,
			     uint32_t *out_len, uint32_t *crc32, int32_t *status);

/* hipdeflate_batch_inflate with hip_inflate_flush's stopping rule */
int hipdeflate_batch_inflate_flush(const uint8_t *in, const uint64_t *in_off,
				   const uint32_t *in_len, uint32_t nblocks,
				   uint8_t *out, const uint64_t *out_off,
				   const uint32_t *out_cap,
				   uint32_t *out_len, uint32_t *crc32, int32_t *status);

/* ---- batch API, device-resident buffers --------------------------------- */
/* Same contracts, every pointer is a DEVICE address (hipMalloc'd, or a torch
 * CUDA tensor's data_ptr()); `stream` is a hipStream_t (NULL = default stream).
 * Asynchronous: returns after enqueueing.  `in` and `out` bases must be
 * 16-byte aligned; fastest when every in_off[i] is too (0xff00 and 0x10000 are). */
int hipdeflate_batch_deflate_dev(const void *in, const void *in_off,
				 const void *in_len, uint32_t nblocks,
				 int level, int frame,
				 void *out, uint64_t out_stride, uint32_t out_cap,
				 void *out_len, void *crc32, void *status,
				 void *stream);
int hipdeflate_batch_inflate_dev(const void *in, const void *in_off,
				 const void *in_len, uint32_t nblocks,
				 void *out, const void *out_off, const void *out_cap,
				 void *out_len, void *crc32, void *status,
				 void *stream);

int hipdeflate_batch_inflate_flush_dev(const void *in, const void *in_off,
				       const void *in_len, uint32_t nblocks,
				       void *out, const void *out_off, const void *out_cap,
				       void *out_len, void *crc32, void *status,
				       void *stream);

/* Gather the variable-length members produced by batch_deflate_dev into one
 * contiguous stream: member i (out_len[i] bytes at slots + i*stride) goes to
 * dst + dst_off[i], where dst_off is the exclusive prefix sum of out_len (plus
 * this rank's base when the stream is sharded across GPUs -- SURVEY.md 8(e)).
 * dst_off is computed on the device by hipdeflate_scan_sizes_dev. */
int hipdeflate_scan_sizes_dev(const void *out_len, uint32_t nblocks,
			      uint64_t base, void *dst_off, void *total,
			      void *stream);
int hipdeflate_compact_dev(const void *slots, uint64_t stride,
			   const void *out_len, const void *dst_off,
			   uint32_t nblocks, void *dst, void *stream);
/* The same gather for ONE RANK'S SPAN of a stream sharded across GPUs: dst_off[] holds
 * offsets in the whole concatenated stream (scan_sizes_dev with base = sum of the lower ranks' totals, the
 * one all_gather of the path), `span` is this rank's buffer and span_base the stream offset of its first
 * byte: member i goes to span + (dst_off[i] - span_base).  The rank then pwrite()s the span at span_base --
 * the in-order writer of applet/7bgzf.c:263-272 without moving payload between GPUs. */
int hipdeflate_compact_span_dev(const void *slots, uint64_t stride,
				const void *out_len, const void *dst_off,
				uint32_t nblocks, void *span, uint64_t span_base,
				void *stream);

/* ---- streaming encoder: the host pipeline either side of the kernels ------------
 * Role of the read / compress / write loop of applet/7bgzf.c:159-293 (7migz.c:130-244)
 * for a stream of fixed-size blocks (the last may be short).  `depth` batches are in
 * flight: the caller fills PINNED input memory directly (no staging copy), H2D copy,
 * kernels and D2H copy of different batches overlap on their own streams, and a
 * result is ONE contiguous run of finished members in block order (the device
 * gathers them), so writing it out is a single write().  Calls on one pipe may come
 * from two threads: one doing input()/submit(), one doing result().
 *
 *   p   = hipdeflate_pipe_open(level, HD_FRAME_BGZF, 0xff00, 4096, 3);
 *   buf = hipdeflate_pipe_input(p, &cap);  n = read(0, buf, cap);  hipdeflate_pipe_submit(p, n);
 *   hipdeflate_pipe_result(p, &data, &nbytes, &nblocks);  write(1, data, nbytes);
 */
typedef struct hipdeflate_pipe hipdeflate_pipe;
/* block_bytes must be a multiple of 16 (0xff00, 0x10000 and b * 1024 are); depth >= 2 */
hipdeflate_pipe *hipdeflate_pipe_open(int level, int frame, uint32_t block_bytes,
				      uint32_t blocks_per_batch, int depth);
/* pinned buffer for the next batch, *cap = block_bytes * blocks_per_batch; waits for
 * a free slot (one whose result has been fetched and released); NULL on error */
uint8_t *hipdeflate_pipe_input(hipdeflate_pipe *p, size_t *cap);
/* enqueue the batch just filled (nbytes <= cap, 0 allowed); returns at once */
int hipdeflate_pipe_submit(hipdeflate_pipe *p, size_t nbytes);
/* the oldest submitted batch: waits for it.  *data stays valid until the next call of
 * hipdeflate_pipe_result on this pipe.  Returns 0; 1 if a block did not fit its slot
 * (cannot happen for BGZF/MiGz block sizes); HD_E_*; HD_E_ARG when nothing is pending */
int hipdeflate_pipe_result(hipdeflate_pipe *p, const uint8_t **data, size_t *nbytes,
			   uint32_t *nblocks);
/* The members of the result last fetched (valid as long as its data): their sizes, their offsets inside the run --
 * the device's size prefix scan, i.e. the compressed offsets a block index needs (bgzip's .gzi, BAM virtual
 * offsets; the role of the index member of applet/7gzinga.c:173-193) -- and the CRC-32 of each block's input.
 * Any of the three may be NULL. */
int hipdeflate_pipe_members(hipdeflate_pipe *p, const uint32_t **out_len, const uint64_t **dst_off,
			    const uint32_t **crc32);
/* BAM / tabix virtual file offset of byte `uoffset` of the block whose member starts at `coffset` */
#define HIPDEFLATE_VOFFSET(coffset, uoffset) (((uint64_t)(coffset) << 16) | (uint64_t)((uoffset) & 0xffff))
void hipdeflate_pipe_close(hipdeflate_pipe *p);
/* the same pipe on entry `index` of the device list: one pipe per device and batches dealt round robin is how
 * hd7bgzf -g N drives N cards from one in-order reader and one in-order writer */
hipdeflate_pipe *hipdeflate_pipe_open_on(int index, int level, int frame, uint32_t block_bytes,
					 uint32_t blocks_per_batch, int depth);

/* ---- streaming decoder: the same pipeline in the other direction -----------------
 * Role of the read / inflate / write loop of applet/7bgzf.c:295-365.  The caller reads
 * compressed bytes into pinned memory, pre-scans the member headers there (the serial
 * BSIZE walk of _read_gz_header, applet/7bgzf.c:81-131) and submits the table; a
 * result is the batch's output as ONE contiguous run (member i at the exclusive
 * prefix sum of out_size[]).  Threading as for hipdeflate_pipe. */
typedef struct hipdeflate_unpipe hipdeflate_unpipe;
hipdeflate_unpipe *hipdeflate_unpipe_open(uint32_t max_members, size_t in_cap, size_t out_cap, int depth);
uint8_t *hipdeflate_unpipe_input(hipdeflate_unpipe *p, size_t *cap);
/* member i: raw DEFLATE at in_off[i] .. +in_len[i] of the buffer (trailing bytes allowed),
 * inflating to exactly out_size[i] bytes (the ISIZE of its trailer); sum(out_size) <= out_cap */
int hipdeflate_unpipe_submit(hipdeflate_unpipe *p, const uint64_t *in_off, const uint32_t *in_len,
			     const uint32_t *out_size, uint32_t nmembers);
/* oldest submitted batch; returns 0, or the first member's non-zero inflate status
 * (1 bad data / 3 does not fit, also used when a member is shorter than out_size), or HD_E_* */
int hipdeflate_unpipe_result(hipdeflate_unpipe *p, const uint8_t **data, size_t *nbytes);
void hipdeflate_unpipe_close(hipdeflate_unpipe *p);
hipdeflate_unpipe *hipdeflate_unpipe_open_on(int index, uint32_t max_members, size_t in_cap, size_t out_cap, int depth);

/* ---- latency contexts: small synchronous batches ----------------------------------
 * For callers that hold a FEW blocks and wait for them: the LD_PRELOAD hook (htslib's worker threads hand over
 * one 0xff00-byte block each), the per-block codecs, a thread-per-block loop like applet/7bgzf.c:159-277 ported
 * as it stands.  A context owns pinned device-visible buffers and a stream: the caller writes block i straight
 * into hipdeflate_lat_input(c, i), hipdeflate_lat_run() codes n blocks (frame | HD_FRAME_LATENCY: several
 * wavefronts per block) and returns when the members are in hipdeflate_lat_output(c, i, ...).  No staging copy,
 * no copy-engine transfer: the kernels read and write the pinned memory themselves.  One thread at a time per
 * context; different contexts run concurrently. */
typedef struct hipdeflate_lat hipdeflate_lat;
hipdeflate_lat *hipdeflate_lat_open(int level, int frame, uint32_t max_blocks, uint32_t max_block_bytes);
/* where block i's input goes (max_block_bytes of pinned memory, 16-byte aligned); NULL if i is out of range */
uint8_t *hipdeflate_lat_input(hipdeflate_lat *c, uint32_t i);
/* code blocks 0..n-1 of in_len[i] bytes; synchronous; 0 if the batch ran (per-block status via _output) */
int hipdeflate_lat_run(hipdeflate_lat *c, const uint32_t *in_len, uint32_t n);
/* member i of the last run: its bytes (pinned, valid until the next run), size, CRC-32 of the input, status */
const uint8_t *hipdeflate_lat_output(hipdeflate_lat *c, uint32_t i, uint32_t *out_len, uint32_t *crc32, int32_t *status);
void hipdeflate_lat_close(hipdeflate_lat *c);
hipdeflate_lat *hipdeflate_lat_open_on(int index, int level, int frame, uint32_t max_blocks, uint32_t max_block_bytes);

/* scratch bytes batch_deflate_dev needs per launch for `level` (0 for level <= 1): the token slabs of the
 * fused kernel plus, for blocks up to 256 KiB (max_block = the slot stride), the tokens and histograms of one
 * sub-batch of the parse + emit kernel pair -- at most 8.25 GiB however large the batch.  The library keeps
 * its own grow-only scratch; this is informational. */
uint64_t hipdeflate_scratch_bytes(uint32_t nblocks, uint32_t max_block, int level);

/* ---- LD_PRELOAD hook ----------------------------------------------------- */
/* Same signature and return values as bgzf_compress.c:39: 0 ok; -1 if *dlen < 26 (28 for the EOF block) or the
 * device is missing; 1 on codec error.  slen == 0 yields the canned 28-byte EOF block.  BGZF_METHOD (parsed once,
 * bgzf_compress.c:53-113: name + trailing digits = level): `hip<level>` is this library's coder (`hip` alone =
 * level 1); UNSET means level 6, as the reference's unset means its zlib at 6 (bgzf_compress.c:54,:102); a name of
 * the reference's table (zlib, libdeflate, igzip, ...) or an unknown one is served by the hip coder at the level
 * the reference would have used for it, with one line on stderr -- a BGZF_METHOD=libdeflate6 left in the
 * environment keeps writing.  There is no CPU codec behind any name.  Calls from concurrent htslib worker threads
 * are micro-batched into latency-mode launches on pinned memory: a batch closes when every caller inside the hook
 * has joined, when nobody has joined for HIPDEFLATE_LINGER_US (8), or after HIPDEFLATE_BATCH_US (60); at most
 * HIPDEFLATE_INFLIGHT (2) batches are on a device at once; batch contexts are spread over the device list.  With up to
 * HIPDEFLATE_MERGE_CALLERS (16) callers a batch that is merely complete waits for the batch on the device (HIPDEFLATE_MERGE_INFLIGHT, 1)
 * and then up to HIPDEFLATE_REJOIN_US (30) for that batch's callers, so that a handful of callers share ONE launch instead of
 * taking turns in two.  Every level has one form behind this call (HD_FRAME_LATENCY above). */
int bgzf_compress(void *dst, size_t *dlen, const void *src, size_t slen, int level);

/* device self-test of the wave primitives (scan, CRC folding); 0 = pass */
int hipdeflate_selftest(void);
/* test entry: the code lengths the device's Huffman construction gives nvec frequency vectors of nsyms
 * (2..288) symbols each under a length limit of maxbits (1..15); lens_out[nvec * nsyms] */
int hipdeflate_test_build_lengths(const uint32_t *freq, uint32_t nvec, uint32_t nsyms, uint32_t maxbits,
				  uint8_t *lens_out);
/* test entry: the schedule of the workgroup levels' throughput form (levels >= 3, launches of 512 blocks and more) --
 * keep = the emit wavefronts a CU keeps resident beside the parse (0..3; 0: none stay, the launches that follow the parses do all
 * the emit work), sub_cap = a cap on the blocks of a sub-batch (0 = none: 17 GiB of records), so that a launch of a few thousand blocks
 * walks the path of a 16 GiB one (two record buffers, the gates between sub-batches).  The bytes do not depend on either.  Process-wide;
 * (3, 0) restores the defaults. */
void hipdeflate_test_beside(int keep, uint32_t sub_cap);

#ifdef __cplusplus
}
#endif
#endif
