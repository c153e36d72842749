// hd_deflate_static.hpp -- level 1: greedy LZ77 + static Huffman, one wavefront
// per block, streaming bit emission (BASELINE config 2).
//
// Replaces, for BGZF_METHOD=hip1, what libdeflate_deflate (lib/zlibutil.c:179)
// -> deflate_compress_fastest (lib/libdeflate/deflate_compress.c:2453-2524)
// -> deflate_flush_block (:1707-2038) do per block on one CPU thread, and the
// CRC-32 of the block (fcrc32, applet/7bgzf.c:269) and the BGZF/MiGz framing
// (applet/7bgzf.c:263-272, applet/7migz.c:224-233) that the applet adds.
//
// The parse is wave64-native (oracle/hd_deflate_twin.c is its serial twin and
// must produce the same bytes):
//   lanes stand on 64 consecutive positions S..S+63; each hashes its 4 bytes,
//   reads the candidate from the 16-bit LDS hash table, then publishes itself
//   (conflicts inside a step are re-written until the largest position holds the
//   slot); candidates are verified against the LDS ring window, 8 bytes per lane;
//   the greedy choice of token starts is a DPP prefix scan over 8-state
//   transition functions (two v_perm_b32 per composition); the rare matches whose
//   8 bytes all agree are extended cooperatively, 64 bytes per ballot; the step's
//   tokens are compacted into an LDS queue, and once 64 wait there one pass turns
//   them into static-Huffman codes, places them with a DPP prefix scan of their
//   bit lengths and ORs them into an LDS staging ring that leaves as 512-byte
//   stores.
//
// HBM traffic per block: input read once (16 B/lane pieces), output written
// once.  LDS per wave: ring 2^WIN_BITS + table (2 B x HD_TABLE_ENTRIES) + 1 KiB staging +
// 640 B token queue = 8848 B at level 1: 7 of the 1280-byte units LDS is granted
// in, 18 waves per CU.
#pragma once
#include <type_traits>
#include "hd_device.hpp"

namespace hd {

constexpr uint32_t HD_HINT_NO_WHOLE = 1;   // no block is short enough to be coded whole
constexpr uint32_t HD_HINT_NO_SEG = 2;     // no block is long enough to be coded in segments

struct DeflateArgs {
	const uint8_t *in;
	const uint64_t *in_off;
	const uint32_t *in_len;
	uint32_t nblocks;
	int frame;
	int level;
	uint8_t *out;
	uint64_t out_stride;
	uint32_t out_cap;
	uint32_t *out_len;
	uint32_t *crc;
	int32_t *status;
	const CrcTables *ct;
	uint8_t *scratch;               // token slabs of the dynamic levels
	uint32_t first;                 // split path: first block of this sub-batch (grid index 0)
	uint32_t count;                 // split path: blocks in this sub-batch
	uint32_t skip_small;            // fused dynamic kernel: leave blocks <= split_max to the split path
	uint32_t split_max;             // split path: largest block it takes (sizes the scratch layout)
	uint32_t *split_ovf;            // split path: per block, 1 = left to the fused kernel (too many tokens)
	uint32_t seg_limit;             // level-1 and fused dynamic kernel: != 0 = leave longer blocks alone (hd_segment.hpp codes them)
	uint32_t seg_bytes;             // ... as segments of this size (HD_SEG_BYTES, or HD_LAT_SEG_BYTES in latency mode)
	// host side only (latency contexts, which see the lengths): launches known to have nothing to do are left out,
	// and the segment table comes ready-made in device-visible memory instead of from k_seg_table
	uint32_t hint;                  // HD_HINT_*
	const uint64_t *host_seg_off;
	const uint32_t *host_seg_len;
	// split path, latency segments of the dynamic levels (HD_LAT_PARTS): != 0 = every block of in_off/in_len is PARSED
	// as this many parts of HD_LAT_PART_BYTES, one wavefront each, and emitted as one DEFLATE block by one
	uint32_t parts = 0;
	// != 0: in_off / in_len are the segment table of a latency-mode launch with this many slots per block; a segment that
	// is not the first of its block is PRIMED with the HD_LAT_PRIME_BYTES before it (the end of its predecessor)
	uint32_t seg_slots = 0;
	// latency contexts: the last launch of a run (k_seg_finish) tells the host it is done by itself -- its last workgroup
	// stores `done_epoch` into `done_flag` (pinned host memory) behind a system-scope fence -- so the caller polls a word
	// instead of going through hipStreamSynchronize.  done_count: a device word the workgroups count themselves off on
	uint32_t *done_flag = nullptr, *done_count = nullptr;
	uint32_t done_epoch = 0;
	// != 0: the records in `scratch` are the workgroup parse's (hd_deflate_wg.hpp wg_layout: a token per byte, a DEFLATE
	// block per HD_WG_SPLIT_MIN bytes); the emit-only kernel reads them so
	uint32_t wg = 0;
	// the workgroup levels, HD_FRAME_LATENCY: a handful of blocks, each wanted back soon -- the member is written by a
	// WORKGROUP (hd_emit_wg.hpp k_emit_wg: the same bytes as the one-wavefront emit kernel, sixteen wavefronts at them)
	uint32_t lat = 0;
	// ... and the PARSE of one block is shared by this many workgroups (k_parse_wg, a.lat launches of a few blocks: 1, 2 or 4):
	// workgroup q replays the table stores of the pieces in front of its range -- hashes and bucket updates only, in the same
	// order, so its table is what the one workgroup's would be there -- and parses its own range; same tokens, a fraction of
	// the time a lone block spends on one CU.  0 = 1
	uint32_t wg_split = 0;
	// ... reading it from here (device memory: block i of the sub-batch at stage_in + i * 64 KiB, put there by k_stage_in)
	// instead of a.in -- which for a latency context is the caller's pinned memory: four readers of a block over PCIe
	// cost more than the shared parse gains
	const uint8_t *stage_in = nullptr;
	// a device word that counts the blocks the workgroup parse gave up on because a turn did not come (WG_SPIN_LIMIT;
	// they are written stored): hipdeflate_stall_count()
	uint32_t *stalls = nullptr;
	// the workgroup levels' throughput form, emit BESIDE the parse (hd_deflate_wg.hpp launch_wg): ready[32 * i] != 0 once block
	// first + i's records are complete (raised by its parse workgroup; the emit wavefront that has the block waits for it; a
	// flag per 128-byte line), arrived counts the emit wavefronts that are resident (the parse is launched behind a gate on it)
	uint32_t *ready = nullptr, *arrived = nullptr, *next = nullptr;     // (next: the counter the blocks are handed out by)
	// ... SPAN: the emit kernel serves the WHOLE launch -- sub-batch k = blocks [k * span_sub, ...) has its records in `scratch` (k even)
	// or `scratch_b` (k odd), its flag at ready[32 * b] for the launch's block b, and emitted[k] counts its members as they are done
	// (the parse of sub-batch k + 2, which overwrites the records, is launched behind a gate on it)
	uint32_t span_sub = 0;
	uint8_t *scratch_b = nullptr;
	uint32_t *emitted = nullptr;
	// ... and in front of that gate a launch of the emit kernel that takes what is left of sub-batch k and of no other (take_sub = k + 1;
	// SPAN hands the blocks out by a counter per sub-batch, next[k]): whatever became of the resident wavefronts, every block of
	// sub-batch k has a live taker before the gate waits.  Should the gate give up all the same (~8 s: a device in trouble), it sets
	// *poison and the parses behind it leave the records alone -- their blocks are written stored and counted as stalls, valid
	// streams, not the twin's bytes
	uint32_t take_sub = 0;
	uint32_t *poison = nullptr;
	uint32_t beside_keep = 3;                    // emit wavefronts a CU keeps (tests: hipdeflate_test_beside(keep = 0): nobody stays)
	// host side only: a WgBeside (hd_deflate_wg.hpp) -- the second stream and the events of that scheme; nullptr = emit behind parse
	void *beside = nullptr;
};

__device__ __forceinline__ uint32_t frame_hdr_bytes(int frame)
{
	return frame == HD_FRAME_BGZF ? 18u : frame == HD_FRAME_MIGZ ? 20u : frame == HD_FRAME_ZLIB ? 2u
	     : frame == HD_FRAME_GZIP ? 10u : 0u;
}

// CRC32 + ISIZE behind the gzip-family members, Adler-32 behind RFC 1950
__device__ __forceinline__ uint32_t frame_trl_bytes(int frame)
{
	return frame == HD_FRAME_ZLIB ? 4u : (frame == HD_FRAME_BGZF || frame == HD_FRAME_MIGZ || frame == HD_FRAME_GZIP) ? 8u : 0u;
}

// 16-bit field `k` of the trailer: CRC32 then ISIZE; the four Adler-32 bytes of an
// RFC 1950 member are left zero here, k_adler32_patch writes them
__device__ __forceinline__ uint32_t frame_trl_field(int frame, uint32_t k, uint32_t crc, uint32_t n)
{
	const uint32_t w = k < 2 ? (frame == HD_FRAME_ZLIB ? 0u : crc) : n;
	return (w >> (16 * (k & 1))) & 0xffff;
}

// HD_FRAME_RAW_FLUSH: room kept for the flush suffix (3 header bits + alignment, 00 00 ff ff)
__device__ __forceinline__ uint32_t frame_sfx_bytes(int frame)
{
	return frame == HD_FRAME_RAW_FLUSH ? 5u : 0u;
}

// dword `j` (0..3) of the container header with its size field zero
__device__ __forceinline__ uint32_t frame_hdr_word(int frame, uint32_t j)
{
	// BGZF / MiGz: 1f 8b 08 04 | 00 00 00 00 | 00 ff XL 00 | S1 S2 SL 00     applet/7bgzf.c:263-265
	// gzip:        1f 8b 08 00 | mtime = 0   | 02 00                          lib/zlibutil.c:380-386
	// zlib:        78 da                                                       lib/zlibutil.c:375-376
	const bool bz = frame == HD_FRAME_BGZF, gz = frame == HD_FRAME_GZIP, zl = frame == HD_FRAME_ZLIB;
	const uint32_t w0 = zl ? 0x0000da78u : gz ? 0x00088b1fu : 0x04088b1fu;
	const uint32_t w2 = zl ? 0u : gz ? 0x00000002u : bz ? 0x0006ff00u : 0x0008ff00u;
	const uint32_t w3 = (zl || gz) ? 0u : bz ? 0x00024342u : 0x00045a4du;
	return j == 0 ? w0 : j == 1 ? 0u : j == 2 ? w2 : w3;
}

// byte `o` of the container header; `sizefield` = BSIZE (u16) or compsize (u32)
__device__ __forceinline__ uint32_t frame_hdr_byte(int frame, uint32_t o, uint32_t sizefield)
{
	const uint32_t w = o < 16 ? frame_hdr_word(frame, o >> 2) : sizefield;
	return (w >> (8 * (o & 3))) & 0xff;
}

// Stored-block form of the whole member, written from scratch (fallback when
// the Huffman stream would not be smaller, deflate_compress.c:1820-1856; the
// guarantee that a 0xff00-byte BGZF block always fits).
__device__ __forceinline__ void write_stored_member(const DeflateArgs &a, uint32_t b, const uint8_t *src, uint32_t n,
						  uint32_t crc, uint32_t lane)
{
	const uint32_t hdr = frame_hdr_bytes(a.frame), trl = frame_trl_bytes(a.frame);
	const bool flush = a.frame == HD_FRAME_RAW_FLUSH;   // no final block; empty stored block 00 | 00 00 ff ff behind
	const uint32_t stored = HD_STORED_SIZE(n);
	const uint32_t total = hdr + stored + trl + (flush ? 5u : 0u);
	uint64_t cap = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
	if (a.frame == HD_FRAME_BGZF && cap > 65536)
		cap = 65536;
	if (total > cap) {
		if (lane == 0) {
			a.out_len[b] = 0;
			if (a.status) a.status[b] = 1;
			if (a.crc) a.crc[b] = crc;
		}
		return;
	}
	const uint32_t sizefield = a.frame == HD_FRAME_BGZF ? total - 1 : stored;
	const uint32_t nblk = n == 0 ? 1u : (n + 65534u) / 65535u;
	uint32_t *dst32 = (uint32_t *)(a.out + (uint64_t)b * a.out_stride);
	for (uint32_t j = lane; j < (total + 3) / 4; j += 64) {
		uint32_t w = 0;
		for (uint32_t k = 0; k < 4; k++) {
			uint32_t o = 4 * j + k, v = 0;
			if (o < hdr) {
				v = frame_hdr_byte(a.frame, o, sizefield);
			} else if (o < hdr + stored) {
				uint32_t q = (o - hdr) / 65540u, r = (o - hdr) - q * 65540u;
				uint32_t left = n - q * 65535u;
				uint32_t bl = left < 65535u ? left : 65535u;
				if (r == 0) v = (q + 1 == nblk && !flush) ? 1u : 0u;
				else if (r == 1) v = bl & 0xff;
				else if (r == 2) v = bl >> 8;
				else if (r == 3) v = ~bl & 0xff;
				else if (r == 4) v = (~bl >> 8) & 0xff;
				else v = src[q * 65535u + r - 5];
			} else if (o < total) {
				uint32_t t = o - hdr - stored;
				v = flush ? (t >= 3 ? 0xffu : 0u) : (frame_trl_field(a.frame, t >> 1, crc, n) >> (8 * (t & 1))) & 0xff;
			}
			w |= v << (8 * k);
		}
		dst32[j] = w;
	}
	if (lane == 0) {
		a.out_len[b] = total;
		if (a.status) a.status[b] = 0;
		if (a.crc) a.crc[b] = crc;
	}
}

// load the 16 bytes of slot (piece, lane); zero beyond n
__device__ __forceinline__ uint4 load_slot(const uint8_t *src, uint32_t n, uint32_t piece, uint32_t lane, bool aligned)
{
	const uint32_t o = piece * HD_PIECE + lane * 16;
	if (aligned && o + 16 <= n)
		return *(const uint4 *)(src + o);
	// ragged tail or unaligned source: byte loads, zero beyond n (static indices
	// only -- a runtime-indexed array would live in scratch)
	uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
	if (o < n) {
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) {
			w0 |= (o + k < n ? (uint32_t)src[o + k] : 0u) << (8 * k);
			w1 |= (o + 4 + k < n ? (uint32_t)src[o + 4 + k] : 0u) << (8 * k);
			w2 |= (o + 8 + k < n ? (uint32_t)src[o + 8 + k] : 0u) << (8 * k);
			w3 |= (o + 12 + k < n ? (uint32_t)src[o + 12 + k] : 0u) << (8 * k);
		}
	}
	return make_uint4(w0, w1, w2, w3);
}

// ---- dynamic levels: parse kernel -> HBM -> emit kernel ------------------------
// Per block of a sub-batch the scratch holds the member's tokens (at most one per
// input byte), the symbol histograms of its DEFLATE blocks (a block closes at the
// first step boundary with >= HD_DYN_BLOCK_TOKENS tokens) and a small record.  The
// parse is this file's kernel with TOK = true: the same steps, but the token queue
// drains into the slab and the histograms instead of into static-Huffman bits.
// hd_deflate_dynamic.hpp's kernel in its emit-only mode turns that into the bytes
// the fused kernel would have written.  The layout is sized by the largest block
// the launch admits (a.split_max: the slot stride bounds it) and holds one token
// per TWO input bytes: compressible data stays well below that (DNA-like 0.2,
// text 0.35 tokens per byte); a block that does not (noise) sets its flag in
// a.split_ovf and is taken by the fused kernel afterwards, as are blocks larger
// than split_max, if any.
//   [ tokens: cap_tok x u32 | ndb, crc, pad, pad | ntok[max_db] | hist[max_db][320] ]
struct SplitLayout {
	uint32_t cap_tok, max_db;
	uint64_t off_rec, off_ntok, off_hist, bytes;
};
// full: room for one token per byte -- the parts of a latency segment (HD_LAT_PARTS), which have no fused kernel behind them
__host__ __device__ inline SplitLayout split_layout(uint32_t max_block, bool full = false)
{
	SplitLayout l;
	// three tokens per four input bytes: text parsed with the short-table levels comes close to one per two
	// (level 4 on the enwik-like set overflowed half its blocks at max_block / 2: 51 GB/s instead of 95)
	l.cap_tok = full ? ((max_block + 64 + 15) & ~15u) : max_block / 4 * 3 + 128;
	l.max_db = l.cap_tok / HD_DYN_BLOCK_TOKENS + 2;
	l.off_rec = (uint64_t)l.cap_tok * 4;
	l.off_ntok = l.off_rec + 16;
	l.off_hist = (l.off_ntok + (uint64_t)l.max_db * 4 + 15) & ~(uint64_t)15;
	l.bytes = l.off_hist + (uint64_t)l.max_db * 320 * 4;
	return l;
}
__device__ __forceinline__ uint8_t *split_block(uint8_t *scratch, uint32_t max_block, uint32_t i)
{
	return scratch + (uint64_t)i * split_layout(max_block).bytes;
}
// the record of part `i` (counted over the sub-batch: segment * parts + part) of a latency segment
__host__ __device__ inline SplitLayout part_layout() { return split_layout(HD_LAT_PART_BYTES, true); }
__device__ __forceinline__ uint8_t *part_block(uint8_t *scratch, uint32_t i)
{
	return scratch + (uint64_t)i * part_layout().bytes;
}

// MINLEN / LAZY / INTRA: the parse variants of the dynamic levels (hd_deflate_dynamic.hpp), used with TOK
// DEEP: the lazy levels' matchfinder (hipdeflate_params.h "LAZY LEVELS"): six-byte key, two positions per bucket, both verified
// PRIMED: the instantiation latency-mode launches use -- parts and priming (HD_LAT_PRIME) are compiled in.  The parse kernels
// always have them; the level-1 kernel of the 16 GiB runs does not (with them in, 0.5 % slower for nothing)
template <int WIN_BITS, int HASH_BITS, bool TOK, int MINLEN = HD_MIN_MATCH, int LAZY = 0, int INTRA = 0, int DEEP = 0, bool PRIMED = TOK>
__global__ __launch_bounds__(64) void k_deflate_static(DeflateArgs a)
{
	static_assert(PRIMED || !TOK, "the parse kernels are always built with the latency-mode paths");
	constexpr uint32_t W = 1u << WIN_BITS;
	constexpr uint32_t W4M = W / 4 - 1;
	// table entries: 16-bit positions, or (DEEP) dword buckets of two -- HS counts 16-bit units either way
	constexpr uint32_t NB = HD_BUCKETS(WIN_BITS, HASH_BITS);
	constexpr uint32_t HS = DEEP ? 2 * NB : HD_TABLE_ENTRIES(WIN_BITS, HASH_BITS);
	static_assert(!DEEP || (TOK && WIN_BITS > 12), "the two-way table belongs to the parse kernels of the lazy levels");
	constexpr uint32_t STG = 256;            // staging ring, dwords
	constexpr uint32_t FLUSH_DW = 128;       // flushed 512 B at a time, 8 B per lane
	constexpr uint32_t TOKQ = 128;           // token queue: < 64 waiting + <= 64 of one step
	// K16: every lane whose first eight bytes agree learns bytes 8..15 in the vector domain too (four more ring
	// dwords per lane, issued ahead of the scan and used behind it), so the scalar extension loop below runs only
	// for matches of 16 bytes and more.  On text 98 % of the "long" matches are 9..15 bytes long, and the parse
	// kernels of the dynamic levels wait on that loop's LDS round trip; the level-1 kernel is bound by its vector
	// instruction count instead and keeps the loop (HD_K16_LEVEL1 to try)
#if defined(HD_K16_ALL)
	constexpr bool K16 = true;
#elif defined(HD_K16_OFF)
	constexpr bool K16 = false;
#elif defined(HD_K16_NOT_L2)
	constexpr bool K16 = TOK && WIN_BITS > 12;
#else
	constexpr bool K16 = TOK;
#endif
	// the ring dwords for it are read a step ahead, in probe(), where five more registers do not cost a wave
	// (every geometry but the level-2 one, which LDS lets run 18 waves per CU: <= 96 VGPRs)
	constexpr bool K16_EARLY = K16 && WIN_BITS > 12;

	// + 16 bytes that mirror the start of the ring, so that the 3 dwords under an
	// unaligned 8-byte read never wrap
	__shared__ __attribute__((aligned(W))) uint32_t ring32[W / 4 + 4];       // W-aligned: ring address = (x & (W - 1)) | base
	// (position + 1) mod 2^16, 0 = empty
	__shared__ __attribute__((aligned(16))) uint16_t table[HS + (DEEP ? 2 : 0)];      // DEEP: + a spare dword for the deferred stores of lanes without a position
	// TOK: no bits are made here, the staging ring's place is taken by the symbol histograms
	// (316 counters: litlen [0,286), offset [286,316).  LDS is granted in 1280-byte units, every byte counts:
	// where it buys another wave per CU -- the level 5-6 geometry -- the counters are 16 bits wide, two to a
	// dword; a DEFLATE block holds fewer than 2^16 tokens, so none can carry into its neighbour.  Elsewhere
	// they stay 32 bits wide: packed ones measured 4 % slower), and the queue has no dump slots
	constexpr uint32_t LDS_REST = (W + 16) + 2 * HS + 4 * TOKQ;      // ring + table + token queue
	constexpr uint32_t WAVES32 = 163840 / (((LDS_REST + 1264 + 1279) / 1280) * 1280);
	constexpr uint32_t WAVES16 = 163840 / (((LDS_REST + 632 + 1279) / 1280) * 1280);
	constexpr bool PACK16 = TOK && WAVES16 > WAVES32;
	__shared__ __attribute__((aligned(16))) uint32_t stage[TOK ? (PACK16 ? 158 : 316) : STG];
	// tokens waiting for the emit pass; [TOKQ, TOKQ + 32) = dump slots of lanes without one (two lanes share
	// a slot: what lands there is never read)
	__shared__ uint32_t tokbuf[TOK ? TOKQ : TOKQ + 32];
	const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)ring32;   // LDS byte address of the ring

	const uint32_t lane = threadIdx.x;
	// PARTS (TOK, a.parts != 0): wavefront = part blockIdx.x % parts of segment a.first + blockIdx.x / parts.  A part behind
	// the first one starts HD_LAT_PRIME_BYTES early: those steps fill table and window and their tokens are dropped
	const bool parted = PRIMED && TOK && a.parts != 0;
	const uint32_t b = a.first + (parted ? blockIdx.x / a.parts : blockIdx.x);
	if (b >= a.nblocks)
		return;
	const ClockStamp clk(TOK ? HD_CLK_PARSE : HD_CLK_STATIC);
	const uint8_t *src = a.in + a.in_off[b];
	uint32_t n = a.in_len[b];
	// PRIMED (latency mode): the wavefront starts `prime` bytes ahead of what it codes and runs those steps as the front of
	// the pipeline only -- table and window are warm when its own bytes begin, and its first matches may reach back
	uint32_t prime = (PRIMED && HD_LAT_SEG_PRIME && a.seg_slots && b % a.seg_slots) ? HD_LAT_PRIME(HD_LAT_PRIME_BYTES, n) : 0u;
	if (parted) {
		const uint32_t o = (blockIdx.x % a.parts) * HD_LAT_PART_BYTES;
		if (o >= n)
			return;                              // the segment ends before this part (the emit wave knows)
		n = n - o < HD_LAT_PART_BYTES ? n - o : HD_LAT_PART_BYTES;
		prime = o ? HD_LAT_PRIME(o, n) : HD_LAT_PRIME(prime, n);
		src += o;
	}
	const uint8_t *const src_own = src;          // what this wavefront codes: [src_own, src_own + n_own)
	const uint32_t n_own = n;
	src -= prime;
	n += prime;
	if (TOK && !parted && n > a.split_max) {
		if (lane == 0)
			a.split_ovf[b] = 1;
		return;                              // the fused kernel takes the large blocks
	}
	if (!TOK && a.seg_limit && n > a.seg_limit)
		return;                              // coded in segments (hd_segment.hpp)
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	uint32_t *dst32 = (uint32_t *)(a.out + (uint64_t)b * a.out_stride);
	const CrcTables *ct = a.ct;

	const uint32_t hdr = frame_hdr_bytes(a.frame), trl = frame_trl_bytes(a.frame);
	uint64_t cap64 = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
	if (a.frame == HD_FRAME_BGZF && cap64 > 65536)
		cap64 = 65536;
	const uint32_t cap = (uint32_t)cap64;
	const uint32_t stored = HD_STORED_SIZE(n_own);
	// the static stream survives only while it stays strictly below the stored
	// size and inside the slot (twin: deflate_static(), `limit`)
	uint32_t limit = stored - 1;
	const bool flush = a.frame == HD_FRAME_RAW_FLUSH;
	const uint32_t sfx = frame_sfx_bytes(a.frame);
	bool use_static = TOK || (a.level >= 1 && cap >= hdr + trl + sfx + 2);   // level 0: stored only
	if (!TOK && use_static && cap - hdr - trl - sfx < limit)
		limit = cap - hdr - trl - sfx;
	const uint32_t limit_bits = limit > 0x1fffffe0u ? 0xffffff00u : 8u * limit;    // (the per-step test in 32 bits; a stream of 2^32 bits is out of the kernel's reach anyway)

	// ---- init LDS -------------------------------------------------------
	for (uint32_t i = lane; i < HS / 8; i += 64)
		((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);
	for (uint32_t i = lane; i < (TOK ? (PACK16 ? 158u : 316u) : STG); i += 64)
		stage[i] = 0;
	if (!TOK && lane < 4 && hdr)
		stage[lane] = frame_hdr_word(a.frame, lane);
	// TOK: where this block's tokens, histograms and record go
	const SplitLayout lay = parted ? part_layout() : split_layout(TOK ? a.split_max : 0);
	uint8_t *const rec = !TOK ? nullptr : parted ? part_block(a.scratch, blockIdx.x) : split_block(a.scratch, a.split_max, blockIdx.x);
	uint32_t *const slab = (uint32_t *)rec;
	uint32_t ntok_slab = 0, db_start = 0, ndb = 0;       // tokens stored; first token / index of the open DEFLATE block
	int32_t db_room = HD_DYN_BLOCK_TOKENS;               // tokens until the open DEFLATE block may close (stored + queued ones counted)

	Fn8Ident fid;
	fid.init(lane);
	HashConsts hk;
	hk.init(HS);
	HashConsts6 hk6;
	hk6.init(NB);
	uint32_t pub_addr = 2u * HS, pub_val = 0;    // DEEP: the last fetch's bucket stores, issued by the next one
	CrcLanes crc;
	// (a primed part's CRC covers [prime, n): the lanes of the priming bytes -- all in the first piece -- fold zeros from
	// state 0, which leaves them at 0, and the register's all-ones start sits on the part's first slot)
	crc.init(lane, n, prime);
	uint32_t filled = 0;                 // ring holds [max(0,filled-W), filled)
	uint4 pre = load_slot(src, n, 0, lane, aligned);

	uint32_t bitpos = 8 * hdr;           // absolute bit position in the slot
	uint32_t flushed = 0;                // dwords already stored to HBM
	const uint32_t paybase = 8 * hdr;

	// OR one token per lane (nbits == 0: none) into the staging ring
	auto put = [&](uint32_t code, uint32_t nbits, uint32_t incl, uint32_t total) {
		// branch-free: a lane without a token ORs zeros; the second dword gets the
		// bits that spill over (zero when the token does not cross a dword)
		const uint32_t bp = bitpos + incl - nbits;
		const uint32_t sh = bp & 31, i = (bp >> 5) & (STG - 1);
		const uint64_t wide = (uint64_t)code << sh;                    // callers pass code = 0 where nbits = 0
		atomicOr(&stage[i], (uint32_t)wide);
		atomicOr(&stage[(i + 1) & (STG - 1)], (uint32_t)(wide >> 32));
		bitpos += total;
	};
	// whenever 128 whole dwords are ready they leave as one 8-byte-per-lane store
	auto flush_ready = [&]() {
		if ((bitpos >> 5) - flushed >= FLUSH_DW) {
			const uint32_t i = (flushed & (STG - 1)) + 2 * lane;   // flushed is a multiple of 128
			const uint2 v = *(const uint2 *)&stage[i];
			*(uint2 *)&stage[i] = make_uint2(0, 0);
			*(uint2 *)&dst32[flushed + 2 * lane] = v;
			flushed += FLUSH_DW;
		}
	};
	auto fill_piece = [&]() {
		const uint32_t piece = filled / HD_PIECE;
		const uint4 v = pre;
		filled += HD_PIECE;
		if (filled < n)
			pre = load_slot(src, n, piece + 1, lane, aligned);
		const uint32_t ro = (piece * HD_PIECE) & (W - 1);
		((uint4 *)ring32)[ro / 16 + lane] = v;
		if (ro == 0 && lane == 0)
			((uint4 *)ring32)[W / 16] = v;          // mirror of ring bytes [0,16)
		crc.fold(ct, piece, piece * HD_PIECE + lane * 16 + 16 <= n && piece * HD_PIECE + lane * 16 >= prime, v);
	};
	// ---- the front of the pipeline -------------------------------------------
	// A step is split in three stages that run one iteration apart, so that no
	// LDS round trip is waited for where it is issued:
	//   fetch(k+2)  own 8 bytes at S + lane, hash, table lookup, publish
	//   probe(k+1)  the three ring dwords under the candidate found by fetch
	//   compute(k)  verify, scan, codes, emit
	// The table holds (position + 1) mod 2^16 in 16 bits (0 = empty): half the
	// LDS of 32-bit entries, which buys occupancy.  Several lanes of a step may
	// publish to one slot in the same store instruction: the highest lane (the
	// largest position) stays, which is the result the CPU twin computes.
	struct Fetched {
		uint32_t v, vh, c;           // own bytes [p,p+4), [p+4,p+8); candidate position + 1 (0 = none)
		uint32_t c2;                 // DEEP: the bucket's older position + 1
	};
	// INNER: every lane of the step has >= 9 bytes left and the block is shorter than
	// 2^16 (all BGZF blocks; all but their last steps), so the end-of-block and the
	// 16-bit wrap-around handling drop out at compile time
	// OWN_AHEAD (the parse kernels, which run two to four waves per SIMD and wait where level 1 issues): the three ring
	// dwords under a lane's own position are read ONE STEP AHEAD of the fetch that hashes them -- a fourth pipeline
	// stage -- so that the table lookup no longer waits for an LDS round trip inside its own step
	// MEASURED AND SWITCHED OFF (round 3, same box): level 6 116.9 -> 112.2 GB/s, level 5 148.8 -> 143.4, level 9 68.2 -> 64.7:
	// at two waves per SIMD every instruction costs its issue slot, the hand-over copies and seven more registers cost
	// more than the round trip they hide.  (And not for the level-2 geometry in any case: 99 VGPRs would cost it a wave.)
	constexpr bool OWN_AHEAD = false;
	struct Own {
		uint32_t w0, w1, w2;
	};
	auto own = [&](uint32_t S_) -> Own {
		const uint32_t *w = &ring32[((S_ + lane) >> 2) & W4M];
		Own o = { w[0], w[1], w[2] };
		return o;
	};
	Own o1 = { 0, 0, 0 };                     // OWN_AHEAD: the dwords under S + 192 + lane
	auto fetch = [&](auto inner_tag, uint32_t S_) -> Fetched {
		constexpr bool INNER = decltype(inner_tag)::value;
		Fetched f;
		const uint32_t p = S_ + lane;
		const uint32_t *w = &ring32[(p >> 2) & W4M];
		uint32_t w0, w1, w2;
		if (OWN_AHEAD && S_ >= 128) {         // (the two fetches of the prologue read their own)
			w0 = o1.w0;
			w1 = o1.w1;
			w2 = o1.w2;
		} else {
			w0 = w[0];
			w1 = w[1];
			w2 = w[2];
		}
		f.v = __builtin_amdgcn_alignbyte(w1, w0, p & 3);
		f.vh = __builtin_amdgcn_alignbyte(w2, w1, p & 3);
		const bool can = INNER || p + (DEEP ? HD_LAZY_KEY_BYTES : HD_MIN_MATCH) <= n;
		// (a lane past the end of the block publishes nothing; it reads slot 0, harmlessly)
		uint32_t e, e2 = 0;
		if (DEEP) {
			// one dword bucket: low half the newest position, high half the one before; every lane stores
			// (what it read << 16) | itself -- { newest before the step, highest lane of the step } stays
			// The store waits for the bucket it is made of: issued here it parked every wave on its own LDS read once
			// per step (SQ_WAIT_ANY 33 % of the deep parse).  It goes out at the head of the NEXT fetch instead --
			// still in front of that step's reads, which is all the order the twin knows: a step's positions are in
			// the table before the next step looks.  (Lanes without a position store to a spare dword behind the table.)
			*(uint32_t *)((uint8_t *)table + pub_addr) = pub_val;
			const uint32_t ha = hash_slot_addr6(f.v, f.vh, hk6);
			const uint32_t eb = *(const uint32_t *)((const uint8_t *)table + (can ? ha : 0u));
			pub_addr = can ? ha : 2u * HS;
			pub_val = (eb << 16) | ((p + 1) & 0xffffu);
			e = eb & 0xffffu;
			e2 = eb >> 16;
		} else {
			const uint32_t ha = hash_slot_addr(f.v, hk);             // byte offset of the entry
			uint16_t *const slot = (uint16_t *)((uint8_t *)table + (can ? ha : 0u));
			const uint16_t mine = (uint16_t)(p + 1);
			e = *slot;
			if (can)
				*slot = mine;
		}
		f.c2 = 0;
		if (INNER) {
			f.c = e;                                         // position + 1 itself: nothing has wrapped
			if (DEEP)
				f.c2 = e2;
		} else {
			// entry -> absolute position + 1 of the latest p' < p with p' + 1 == e (mod 2^16)
			const uint32_t back = (p + 1 - e) & 0xffffu;         // 0: an entry exactly 2^16 back, i.e. stale
			f.c = (can && e && back) ? p + 1 - back : 0u;
			if (DEEP) {
				const uint32_t back2 = (p + 1 - e2) & 0xffffu;
				f.c2 = (can && e2 && back2) ? p + 1 - back2 : 0u;
			}
		}
		if (INTRA) {
			// a nearer occurrence inside the step replaces the table's candidate
			const uint32_t d = intra_step_distance<INTRA>(f.v, lane);
			f.c = (can && d) ? p + 1 - d : f.c;
		}
		// Lanes of this step whose four bytes hash alike stored to one entry in that single ds_write_b16:
		// on gfx950 the highest lane's data stays -- the largest position, which is what the CPU twin
		// keeps.  hipdeflate_selftest() checks that property of the LDS on the device (hd_selftest.hip,
		// k_selftest_lds_order), every kernel-vs-twin test leans on it.  (A read-back loop that re-wrote
		// until the slot held the maximum stood here in round 1; the compiler forwarded each lane's own
		// store to its re-read, so it never ran -- and never had to.)
		return f;
	};
	struct Probed {
		uint32_t c0, c1, c2;
		uint32_t c3, c4, p2, p3, p4;     // K16: the candidate's and the lane's own bytes up to 16 (+ alignment)
		uint32_t d0, d1, d2, d3, d4;     // DEEP: the five dwords under the bucket's older position
	};
	auto probe = [&](uint32_t c, uint32_t p, uint32_t cB = 0) -> Probed {
		Probed q;
		if (DEEP) {
			const uint32_t *wb = &ring32[((cB - 1) >> 2) & W4M];
			q.d0 = wb[0];
			q.d1 = wb[1];
			q.d2 = wb[2];
			q.d3 = wb[3];
			q.d4 = wb[4];
		} else {
			q.d0 = q.d1 = q.d2 = q.d3 = q.d4 = 0;
		}
		const uint32_t *w = &ring32[((c - 1) >> 2) & W4M];
		q.c0 = w[0];
		q.c1 = w[1];
		q.c2 = w[2];
		if (K16_EARLY) {
			// (+ 16 mirrored bytes behind the ring: dwords [0,5) of an index never wrap)
			const uint32_t *wp = &ring32[(p >> 2) & W4M];
			q.c3 = w[3];
			q.c4 = w[4];
			q.p2 = wp[2];
			q.p3 = wp[3];
			q.p4 = wp[4];
		} else {
			q.c3 = q.c4 = q.p2 = q.p3 = q.p4 = 0;
		}
		return q;
	};

	// ---- the back of the pipeline: codes + bit packing for up to 64 queued tokens
	// (straight-line: both forms computed, one selected); false = the static
	// stream no longer fits under `limit`
	uint32_t qhead = 0, qtail = 0;
	auto count_symbol = [&](uint32_t c) {                             // TOK only
		if (PACK16)
			atomicAdd(&stage[c >> 1], 1u << (16 * (c & 1)));
		else
			atomicAdd(&stage[c], 1u);
	};
	auto emit_tokens = [&](uint32_t count) -> bool {
		const uint32_t t = tokbuf[(qhead + lane) & (TOKQ - 1)];
		qhead += count;
		if (!use_static)
			return false;                            // given up earlier in this group of steps: nothing more is made
		if (TOK) {
			// queued tokens -> slab (one coalesced 4 B/lane store) + symbol histograms
			if (ntok_slab + count > lay.cap_tok)
				return use_static = false;           // more tokens than the slab holds: the fused kernel's block
			if (lane < count) {
				slab[ntok_slab + lane] = t;
				if (t & HD_TOKEN_MATCH) {
					uint32_t ls, leb, lev, ds, deb, dev;
					len_slot(((t >> 16) & 0xff) + 3, ls, leb, lev);
					off_slot((t & 0xffff) + 1, ds, deb, dev);
					count_symbol(257 + ls);
					count_symbol(286 + ds);
				} else {
					count_symbol(t & 0xff);
				}
			}
			ntok_slab += count;
			return true;
		}
		// literal / length: one table load (CrcTables::SL); offset: slot arithmetic, 5-bit code + extra bits
		const uint64_t mmask = __ballot((int32_t)t < 0);
		const uint64_t vmask = count >= 64 ? ~0ull : (1ull << (count & 63)) - 1;
		const uint32_t e0 = ct->SL[sel(mmask, (t >> 16) & 0x1ff, t & 0xff)];
		uint32_t ds, deb, dev;
		off_slot((t & 0xffff) + 1, ds, deb, dev);
		const uint32_t dpart = (__brev(ds) >> 27) | (dev << 5);
		const uint32_t nb0 = e0 >> 16;
		const uint32_t code = sel(vmask, sel(mmask, (e0 & 0xffff) | (dpart << nb0), e0 & 0xffff), 0u);
		const uint32_t nbits = sel(vmask, sel(mmask, nb0 + 5 + deb, nb0), 0u);
		const uint32_t incl = wave_incl_scan<true>(nbits);
		const uint32_t total = readlane(incl, 63);
		// (in 32 bits: a 64-bit compare of two uniform values is a VECTOR compare on this machine -- v_mov_b64 + v_cmp_gt_u64 per step.
		// limit_bits <= 0xffffff00 and what has passed this test is <= limit_bits, so the sum cannot wrap)
		if ((bitpos - paybase) + total + 7 > limit_bits)
			return use_static = false;
		put(code, nbits, incl, total);
		flush_ready();
		return true;
	};

	// BFINAL = 1 (0 in flush form), BTYPE = 01
	if (!TOK)
		put(lane == 0 ? (flush ? 2u : 3u) : 0u, lane == 0 ? 3u : 0u, 3u, 3u);

	Fetched f0 = { 0, 0, 0, 0 }, f1 = { 0, 0, 0, 0 };
	Probed q0 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	if (use_static && n) {
		fill_piece();
		f0 = fetch(std::false_type{}, 0);
		q0 = probe(f0.c, lane, f0.c2);
		f1 = fetch(std::false_type{}, 64);
		if (OWN_AHEAD)
			o1 = own(128);
	}
	uint32_t carry = 0;                  // leading positions covered by the last match
	// one step; false = the static stream was abandoned
	// NOFILL: the caller has refilled the ring for this step (the 16-step groups of the main loop)
	auto step = [&](auto inner_tag, auto nofill_tag, uint32_t S) -> bool {
		constexpr bool INNER = decltype(inner_tag)::value;
		constexpr bool NOFILL = decltype(nofill_tag)::value;
		if (!NOFILL && filled < n && filled < S + HD_LOOKAHEAD)
			fill_piece();
		const uint32_t lo = filled > W ? filled - W : 0;
		const uint32_t lanes = INNER ? 64u : (n - S < 64 ? n - S : 64);

		// stage 2 of step k+1 and stage 1 of step k+2 go out first
		const Fetched fc = f0;
		const Probed qc = q0;
		f0 = f1;
		q0 = probe(f1.c, S + 64 + lane, f1.c2);   // harmless beyond n: every index is masked into the ring
		f1 = fetch(inner_tag, S + 128);
		if (OWN_AHEAD)
			o1 = own(S + 192);                 // (every index is masked into the ring; bytes beyond the block never count)

		// ---- 3. verify the candidate + first 8 bytes of its length ---------
		// Conditions live as 64-bit lane masks in scalar registers (hd_device.hpp "lane masks"): each is the
		// ballot of ONE compare, they are combined by s_and / s_andn2 and reach the lanes again through sel().
		const uint64_t lanem = (INNER || lanes == 64) ? ~0ull : (1ull << (lanes & 63)) - 1;   // lanes inside the block
		const uint32_t p = S + lane;
		const uint32_t cv0 = fc.v, cvh0 = fc.vh;
		const uint32_t room = n - p;                  // >= 4 where a match can start
		uint32_t c = fc.c, cp = c - 1;
		uint32_t cv = __builtin_amdgcn_alignbyte(qc.c1, qc.c0, cp & 3);
		uint32_t cvh = __builtin_amdgcn_alignbyte(qc.c2, qc.c1, cp & 3);
		uint32_t x = cvh ^ cvh0;
		uint64_t xm = __ballot(cvh == cvh0);                            // all eight bytes agree
		uint32_t len8 = sel(xm, 8u, 4u + ((uint32_t)__builtin_ctz(x) >> 3));     // (x == 0: not selected)
		// candidate inside the window (c == 0, no candidate, makes cp = -1: one signed compare covers both;
		// this kernel never sees a block of 2 GiB), its four bytes equal
		uint64_t okm = __ballot((int32_t)cp >= (int32_t)lo) & __ballot(cv == cv0);
		// bytes 8..15 of the candidate against the lane's own (valid where the first eight agree): K16, and DEEP's choice
		uint32_t qc3 = qc.c3, qc4 = qc.c4, qc2 = qc.c2;
		auto tail16 = [&](uint32_t w2, uint32_t w3, uint32_t w4, uint32_t cpos) -> uint32_t {
			const uint32_t xa = __builtin_amdgcn_alignbyte(qc.p3, qc.p2, p & 3) ^ __builtin_amdgcn_alignbyte(w3, w2, cpos & 3);   // bytes 8..11
			const uint32_t xb = __builtin_amdgcn_alignbyte(qc.p4, qc.p3, p & 3) ^ __builtin_amdgcn_alignbyte(w4, w3, cpos & 3);   // bytes 12..15
			// (v_ffbl_b32 of 0 is -1: >> 3 and min 4 turn it into "all four agree"; ka >> 2 is 1 exactly then)
			uint32_t fa, fb;
			asm("v_ffbl_b32 %0, %1" : "=v"(fa) : "v"(xa));
			asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(xb));
			const uint32_t ka = (fa >> 3) < 4 ? (fa >> 3) : 4u, kb = (fb >> 3) < 4 ? (fb >> 3) : 4u;
			return 8 + ka + (ka >> 2) * kb;
		};
		uint32_t len16_deep = 8;
		if (DEEP) {
			// the bucket's older position, verified the same way; it is taken only when it is strictly longer over
			// 16 bytes (cut to the room that is left, as the twin's prefix lengths are)
			const uint32_t cB = fc.c2, cpB = cB - 1;
			const uint32_t cvB = __builtin_amdgcn_alignbyte(qc.d1, qc.d0, cpB & 3);
			const uint32_t cvhB = __builtin_amdgcn_alignbyte(qc.d2, qc.d1, cpB & 3);
			const uint32_t xB = cvhB ^ cvh0;
			const uint64_t xmB = __ballot(cvhB == cvh0);
			const uint32_t len8B = sel(xmB, 8u, 4u + ((uint32_t)__builtin_ctz(xB) >> 3));
			const uint64_t okB = __ballot((int32_t)cpB >= (int32_t)lo) & __ballot(cvB == cv0);
			const uint32_t l16A = tail16(qc.c2, qc.c3, qc.c4, cp), l16B = tail16(qc.d2, qc.d3, qc.d4, cpB);
			uint32_t LA = sel(okm, sel(xm, l16A, len8), 0u), LB = sel(okB, sel(xmB, l16B, len8B), 0u);
			if (!INNER) {
				LA = LA < room ? LA : room;
				LB = LB < room ? LB : room;
			}
			const uint64_t chB = __ballot(LB > LA);
			c = sel(chB, cB, c);
			cp = c - 1;
			len8 = sel(chB, len8B, len8);
			len16_deep = sel(chB, l16B, l16A);
			qc2 = sel(chB, qc.d2, qc.c2);
			qc3 = sel(chB, qc.d3, qc.c3);
			qc4 = sel(chB, qc.d4, qc.c4);
			xm = (xm & ~chB) | (xmB & chB);
			okm = (okm & ~chB) | (okB & chB);
		}
		const uint32_t mylen = INNER ? len8 : (len8 < room ? len8 : room);     // <= 8 until a capped match is extended
		if (!INNER)
			okm &= __ballot(p + (DEEP ? HD_LAZY_KEY_BYTES : HD_MIN_MATCH) <= n);
		if (MINLEN > HD_MIN_MATCH)
			okm &= __ballot(mylen >= (uint32_t)MINLEN);
		if (LAZY) {
			// a candidate steps aside when its right neighbour's 8-byte length is longer
			const uint32_t l8 = sel(okm, mylen, 0u);
			const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)l8, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
			okm &= ~(__ballot(nx > l8) & (lanem >> 1));                      // (lane + 1 < lanes)
		}

		if (carry >= lanes) {                // the whole step lies inside the last match
			carry -= lanes;
			return true;
		}

		// ---- 4. greedy resolution as a wave prefix scan ---------------------
		// The greedy parse is a little automaton walking the lanes: its state
		// r = "lanes still covered by the current match"; a lane with r == 0
		// starts a token and sets r = len - 1 (0 for a literal), otherwise
		// r -= 1.  Each lane's transition f_l : {0..7} -> {0..7} is eight bytes;
		// composing two of them is two v_perm_b32 table lookups, so the state
		// entering every lane comes out of a 6-stage DPP scan -- no per-match
		// serial loop (DNA-like data has ~12 matches per 64 bytes).  Matches
		// whose first 8 bytes all agree ("capped") are rare; when the scan
		// takes one it is extended cooperatively and the scan is redone for
		// the lanes behind it.
		const uint64_t capmask = INNER ? (okm & xm) : (okm & xm & __ballot(room > 8));
		const uint32_t jump8 = sel(okm, mylen, 1u);                  // token length as the scan sees it
		uint32_t lenv = jump8;                                       // ... and with capped matches extended
		const uint64_t livem = ~0ull << carry;                       // lanes the last match does not cover (carry < 64)
		// (token words are prepared here, ahead of the scan: independent work for the wait states between its
		// DPP stages)  match: HD_TOKEN_MATCH_TAG | (len - 3) << 16 | (dist - 1), dist - 1 = p - c
		const uint32_t mw_base = (p + (HD_TOKEN_MATCH_TAG - (3u << 16))) - c;
		const uint32_t lit = cv0 & 0xff;
		// K16: length over 16 bytes (valid on the capped lanes).  (The level-2 geometry reads the dwords here and is
		// bound by its instruction count: it skips all of this in a step without a capped lane -- a quarter of the
		// steps on FASTQ-like data)
		uint32_t len16 = 8;
		const bool k16_step = K16 && (K16_EARLY || capmask != 0);
		if (DEEP) {
			len16 = INNER ? len16_deep : (len16_deep < room ? len16_deep : room);
		} else if (k16_step) {
			const uint32_t *wp = &ring32[(p >> 2) & W4M], *wc = &ring32[(cp >> 2) & W4M];
			const uint32_t c2 = qc2;
			const uint32_t p2 = K16_EARLY ? qc.p2 : wp[2], p3 = K16_EARLY ? qc.p3 : wp[3], p4 = K16_EARLY ? qc.p4 : wp[4];
			const uint32_t c3 = K16_EARLY ? qc3 : wc[3], c4 = K16_EARLY ? qc4 : wc[4];
			const uint32_t xa = __builtin_amdgcn_alignbyte(p3, p2, p & 3) ^ __builtin_amdgcn_alignbyte(c3, c2, cp & 3);   // bytes 8..11
			const uint32_t xb = __builtin_amdgcn_alignbyte(p4, p3, p & 3) ^ __builtin_amdgcn_alignbyte(c4, c3, cp & 3);   // bytes 12..15
			// (v_ffbl_b32 of 0 is -1: >> 3 and min 4 turn it into "all four agree"; ka >> 2 is 1 exactly then)
			uint32_t fa, fb;
			asm("v_ffbl_b32 %0, %1" : "=v"(fa) : "v"(xa));
			asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(xb));
			const uint32_t ka = (fa >> 3) < 4 ? (fa >> 3) : 4u, kb = (fb >> 3) < 4 ? (fb >> 3) : 4u;
			len16 = 8 + ka + (ka >> 2) * kb;
			if (!INNER)
				len16 = len16 < room ? len16 : room;
		}
		// K16, continuation lanes.  A capped match of 9..15 bytes at lane m is taken apart for the automaton: m
		// jumps 8, and lane m + 8 -- its "continuation lane" -- jumps the k = len - 8 bytes that are left, whatever
		// candidate that lane has itself.  A parse that takes m arrives at m + 8 with state 0, walks on k lanes and
		// is where the match of 8 + k bytes would have put it: no scalar work at all.  The continuation lane is a
		// token start for the scan only (`v` below: not queued; the token of m carries the whole length).  What
		// is left for the scalar loop: matches of 16 bytes and more (k8m), and continuation lanes the parse reached
		// WITHOUT taking their parent (the parent lies under an earlier token): there the lane's own candidate
		// counts, and the chain is re-threaded from it.  On text that is one event in 12 steps instead of 2.5 per step.
		//   A     parents: capped, 9..15 bytes, not themselves the continuation lane of a parent (stride-8 chains
		//         alternate: A[l] = capk[l] & ~A[l - 8])
		//   contm continuation lanes (A << 8; a parent in lanes 56..63 has none: its token simply ends in the next step)
		//   jumpA what the automaton sees, jumpW the true length of a token that starts at the lane (16+: 8 for now)
		uint64_t k8m = 0, par = 0, contm = 0;
		uint32_t jumpA = jump8, jumpW = jump8;
		if (k16_step) {
			k8m = capmask & __ballot(len16 == 16);
			const uint64_t capk = capmask & ~k8m & __ballot(len16 != 8);
			par = capk;
			if (capk & (capk << 8)) {
#pragma unroll
				for (int i = 0; i < 7; i++)
					par = capk & ~(par << 8);
			}
			contm = (par << 8) & lanem;
			const uint32_t kc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane - 8) << 2), (int)(len16 - 8));
			jumpW = sel(capk, len16, jump8);
			jumpA = sel(contm, kc, jump8);
			lenv = jumpW;
		}
		uint64_t starts;
		{
			// fn8_make: {a, 0, 1, 2 | 3, 4, 5, 6} with a = jump - 1, the identity on covered lanes
			Fn8 f;
			f.lo = sel(livem, jumpA + 0x0200ffffu, 0x03020100u);
			f.hi = sel(livem, 0x06050403u, 0x07060504u);
			const Fn8 w = fn8_scan(f, fid);
			// state entering lane l = (f_{l-1} o ... o f_0)(0): byte 0 of lane l-1 (0 enters lane 0)
			const uint32_t sin = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w.lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
			starts = __ballot((sin & 0xff) == 0) & livem;
		}
		// Capped matches the scan took: extend each (left to right) to its true
		// length, drop the token starts it now covers, and re-thread the chain
		// behind it.  Two parses that start a token on the same lane coincide
		// from there on, so the walk stops at the first old start it lands on
		// (a few hops) instead of re-scanning the wave.
		uint64_t v = 0;                      // K16: continuation lanes whose parent the parse took
		if (K16) {
			uint64_t pend = ~0ull;           // lanes behind the last event
			for (;;) {
				v = starts & contm & ((starts & par) << 8);
				const uint64_t ev = starts & (k8m | contm) & ~v & pend;
				if (!ev)
					break;
				const uint32_t m = (uint32_t)__ffsll((unsigned long long)ev) - 1;
				uint32_t len;
				if ((k8m >> m) & 1) {
					// 16 bytes and more: extended as below, from byte 16 on
					const uint32_t pm = S + m;
					const uint32_t dm = pm + 1 - readlane(c, m);
					const uint32_t maxlen = n - pm < HD_MAX_MATCH ? n - pm : HD_MAX_MATCH;
					uint32_t t0, t1, k;
					len = 16;
					asm volatile("Lhd_ext_%=:\n\t"
					     "v_add_u32 %[t0], %[len], %[vb]\n\t"
					     "v_add_u32 %[t1], %[ndm], %[t0]\n\t"
					     "v_and_or_b32 %[t0], %[t0], %[msk], %[rb]\n\t"
					     "v_and_or_b32 %[t1], %[t1], %[msk], %[rb]\n\t"
					     "ds_read_u8 %[t0], %[t0]\n\t"
					     "ds_read_u8 %[t1], %[t1]\n\t"
					     "s_waitcnt lgkmcnt(0)\n\t"
					     "v_cmp_ne_u16 vcc, %[t0], %[t1]\n\t"
					     "s_cbranch_vccnz Lhd_ext_hit_%=\n\t"
					     "s_add_u32 %[len], %[len], 64\n\t"
					     "s_cmp_lt_u32 %[len], %[maxlen]\n\t"
					     "s_cbranch_scc1 Lhd_ext_%=\n\t"
					     "s_branch Lhd_ext_done_%=\n"
					     "Lhd_ext_hit_%=:\n\t"
					     "s_ff1_i32_b64 %[k], vcc\n\t"
					     "s_add_u32 %[len], %[len], %[k]\n"
					     "Lhd_ext_done_%=:"
					     : [len] "+s"(len), [t0] "=&v"(t0), [t1] "=&v"(t1), [k] "=&s"(k)
					     : [vb] "v"(pm + lane), [ndm] "s"(0u - dm), [msk] "s"(W - 1), [rb] "v"(ring_lds), [maxlen] "s"(maxlen)
					     : "vcc", "scc", "memory");
					len = len < maxlen ? len : maxlen;
					// (v_writelane_b32 with the lane select in M0: the one form that may name two scalar operands on
					// gfx9.  Nothing else in this kernel uses M0 -- LDS instructions do not need it here)
					asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(lenv) : "s"(len), "s"(m));
				} else {
					len = readlane(jumpW, m);            // a continuation lane on its own: its own candidate (or a literal)
				}
				// re-thread behind the token of m.  The old chain is good again from its first start that is not a
				// continuation lane (those behave differently when they are reached from elsewhere)
				uint64_t fresh = 0;
				// (the merge of an asm result and a v_readlane counts as divergent for the compiler: pinned)
				uint32_t xq = m + (uint32_t)__builtin_amdgcn_readfirstlane((int)len), hop;
				asm volatile("s_cmp_gt_u32 %0, 63\n\t"
					     "s_cbranch_scc1 Lhd_rethread_done_%=\n"
					     "Lhd_rethread_%=:\n\t"
					     "s_bitcmp1_b64 %3, %0\n\t"
					     "s_cbranch_scc1 Lhd_rethread_done_%=\n\t"
					     "s_bitset1_b64 %1, %0\n\t"
					     "v_readlane_b32 %2, %4, %0\n\t"
					     "s_add_u32 %0, %0, %2\n\t"
					     "s_cmp_lt_u32 %0, 64\n\t"
					     "s_cbranch_scc1 Lhd_rethread_%=\n"
					     "Lhd_rethread_done_%=:"
					     : "+s"(xq), "+s"(fresh), "=&s"(hop)
					     : "s"(starts & ~contm), "v"(jumpW)
					     : "scc");
				const uint32_t xe = xq < 64 ? xq : 64;
				uint64_t gone;                           // lanes (m, xe): xe - m - 1 may be 0 here (a literal at m)
				asm("s_bfm_b64 %0, %1, %2" : "=s"(gone) : "s"(xe - m - 1), "s"(m + 1));
				starts = (starts & ~gone) | fresh;
				pend = ~1ull << m;
			}
		} else {
			uint64_t cm = starts & capmask;
			while (cm) {
				const uint32_t m = (uint32_t)__ffsll((unsigned long long)cm) - 1;
				const uint32_t pm = S + m;
				const uint32_t dm = pm + 1 - readlane(c, m);
				const uint32_t maxlen = n - pm < HD_MAX_MATCH ? n - pm : HD_MAX_MATCH;
				uint32_t len = 8;
				{
					// 64 bytes per pass, every index masked into the ring (lanes past maxlen may read too: the length
					// is cut to maxlen behind the loop).  Written out: left to the compiler the two exits become 11
					// scalar instructions of cselect per pass; here 3 (mismatch found) or 4 (another pass).
					uint32_t t0, t1, k;
					asm volatile("Lhd_ext_%=:\n\t"
						     "v_add_u32 %[t0], %[len], %[vb]\n\t"
						     "v_add_u32 %[t1], %[ndm], %[t0]\n\t"
						     "v_and_or_b32 %[t0], %[t0], %[msk], %[rb]\n\t"
						     "v_and_or_b32 %[t1], %[t1], %[msk], %[rb]\n\t"
						     "ds_read_u8 %[t0], %[t0]\n\t"
						     "ds_read_u8 %[t1], %[t1]\n\t"
						     "s_waitcnt lgkmcnt(0)\n\t"
						     "v_cmp_ne_u16 vcc, %[t0], %[t1]\n\t"
						     "s_cbranch_vccnz Lhd_ext_hit_%=\n\t"
						     "s_add_u32 %[len], %[len], 64\n\t"
						     "s_cmp_lt_u32 %[len], %[maxlen]\n\t"
						     "s_cbranch_scc1 Lhd_ext_%=\n\t"
						     "s_branch Lhd_ext_done_%=\n"
						     "Lhd_ext_hit_%=:\n\t"
						     "s_ff1_i32_b64 %[k], vcc\n\t"
						     "s_add_u32 %[len], %[len], %[k]\n"
						     "Lhd_ext_done_%=:"
						     : [len] "+s"(len), [t0] "=&v"(t0), [t1] "=&v"(t1), [k] "=&s"(k)
						     : [vb] "v"(pm + lane), [ndm] "s"(0u - dm), [msk] "s"(W - 1), [rb] "v"(ring_lds), [maxlen] "s"(maxlen)
						     : "vcc", "scc", "memory");
				}
				len = len < maxlen ? len : maxlen;
				if (len > 8) {
					// (v_writelane_b32 with the lane select in M0: the one form that may name two scalar operands on
					// gfx9.  Nothing else in this kernel uses M0 -- LDS instructions do not need it here)
					asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(lenv) : "s"(len), "s"(m));
					const uint32_t q = m + len;                   // first lane behind the match
					// the walk, written out (the compiler spends ~11 scalar instructions per hop on it; here 4 + 2
					// branches + one v_readlane.  The lane selects of v_readlane / s_bitcmp1 / s_bitset1 take the low
					// 6 bits and xq < 64 is tested first; an SGPR written by the SALU needs no wait states before
					// v_readlane uses it as lane select)
					uint64_t fresh = 0;
					uint32_t xq = q, hop;
					asm volatile("s_cmp_gt_u32 %0, 63\n\t"
						     "s_cbranch_scc1 Lhd_rethread_done_%=\n"
						     "Lhd_rethread_%=:\n\t"
						     "s_bitcmp1_b64 %3, %0\n\t"
						     "s_cbranch_scc1 Lhd_rethread_done_%=\n\t"
						     "s_bitset1_b64 %1, %0\n\t"
						     "v_readlane_b32 %2, %4, %0\n\t"
						     "s_add_u32 %0, %0, %2\n\t"
						     "s_cmp_lt_u32 %0, 64\n\t"
						     "s_cbranch_scc1 Lhd_rethread_%=\n"
						     "Lhd_rethread_done_%=:"
						     : "+s"(xq), "+s"(fresh), "=&s"(hop)
						     : "s"(starts), "v"(jump8)
						     : "scc");
					// the old parse holds from lane xq on: its starts in (m, xq) go (one s_bfm_b64: xq - m - 1 >= 8 ones
					// from bit m + 1; xq >= 64 makes the run end at lane 63, m == 63 makes it empty)
					const uint32_t xe = xq < 64 ? xq : 64;
					uint64_t gone;
					asm("s_bfm_b64 %0, %1, %2" : "=s"(gone) : "s"(xe - m - 1), "s"(m + 1));
					starts = (starts & ~gone) | fresh;
				}
				cm = starts & capmask & (~1ull << m);             // lanes > m
			}
		}
		// coverage behind the last token of the step
		// (the last REAL start: a continuation lane behind it belongs to its token)
		const uint32_t last = 63 - (uint32_t)__clzll((long long)(starts & ~v));      // != 0: carry < lanes, and the first start is real
		const uint32_t E = last + readlane(lenv, last);
		const uint64_t tm = starts & lanem & ~v;           // tokens: matches + literals inside the block
		const uint64_t mm = tm & okm;                      // matches
		// max(E, 64) - 64 in scalar registers (written in C the compiler makes a v_sub ... clamp + v_readfirstlane of
		// it); tail step: matches are clipped to n
		if (INNER || lanes == 64)
			asm("s_max_u32 %0, %1, 64\n\ts_sub_u32 %0, %0, 64" : "=s"(carry) : "s"(E) : "scc");
		else
			carry = 0;

		// ---- 5. queue the step's tokens in position order -------------------
		// token word: literal byte, or HD_TOKEN_MATCH_TAG | (len - 3) << 16 | (dist - 1), dist - 1 = p - c.
		// The code generation and the bit packing below cost the same for 1 or 64
		// tokens, and DNA-like input yields only ~14 tokens per step: they wait in
		// a small LDS ring until 64 are there.
		{
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0));
			const uint32_t tw = sel(mm, (lenv << 16) + mw_base, lit);
			const uint32_t qslot = (qtail + rank) & (TOKQ - 1);
			if (TOK) {
				if ((tm >> lane) & 1)
					tokbuf[qslot] = tw;
			} else {
				tokbuf[sel(tm, qslot, TOKQ + (lane & 31))] = tw;
			}
			qtail += (uint32_t)__popcll(tm);
			if (TOK)
				db_room -= (int32_t)__popcll(tm);
		}
		// (a failed pass -- the stream would pass `limit`, or the slab is full -- clears use_static and the
		// passes after it do nothing: the steps of a group need no exits between them)
		if (qtail - qhead >= 64)
			emit_tokens(64);
		return use_static;
	};
	// fetch runs two steps ahead: a step is INNER when the lanes of step S + 128 still have 9 bytes
	// (<= 2^16: the last position that enters the table is n - 4 (n - 6 at the lazy levels), so position + 1 still fits the
	// 16-bit entries of a block of exactly 0x10000 bytes -- the reference's single-thread block size, applet/7bgzf.c:146-147.
	// Such blocks had been running through the general steps: 244 GB/s against 323 for 0xff00-byte blocks, which rounds 1-2
	// took for HBM channel aliasing; a start stagger of the waves, tried on that theory, only cost time.)
	const bool small = n <= 65536;
	// TOK: the open DEFLATE block's histograms leave for HBM and start again from zero
	auto close_deflate_block = [&]() {
		uint32_t *h = (uint32_t *)(rec + lay.off_hist) + ndb * 320;
		// HBM layout: u32 litlen [0,288), offset [288,320)
		for (uint32_t i = lane; i < 320; i += 64) {
			const uint32_t j = i < 286 ? i : i - 2;          // counter of litlen i / offset i - 288
			const bool used = i < 286 || (i >= 288 && i < 318);
			h[i] = !used ? 0u : PACK16 ? (stage[j >> 1] >> (16 * (j & 1))) & 0xffffu : stage[j];
		}
		for (uint32_t i = lane; i < (PACK16 ? 158u : 316u); i += 64)
			stage[i] = 0;
		if (lane == 0)
			((uint32_t *)(rec + lay.off_ntok))[ndb] = ntok_slab - db_start;
		db_start = ntok_slab;
		db_room = HD_DYN_BLOCK_TOKENS;
		ndb++;
	};
	// a DEFLATE block closes at the first step boundary with >= 32768 tokens (as the fused kernel)
	// (db_room == HD_DYN_BLOCK_TOKENS - (ntok_slab + queued - db_start); an INNER step is never the last one)
	auto step_boundary = [&](auto inner_tag, uint32_t S) -> bool {
		constexpr bool INNER = decltype(inner_tag)::value;
		if (TOK && db_room <= 0 && use_static && (INNER || S + 64 < n)) {
			if (qtail != qhead && !emit_tokens(qtail - qhead))
				return false;
			close_deflate_block();
		}
		return use_static;
	};
	// The ring is refilled one 1 KiB piece at a time, in the step that would otherwise run out of lookahead:
	// S = 704, 1728, ... (filled == S + 320).  From there the next 16 steps need no refill, and while all
	// of them are INNER steps they run as one GROUP: the refill once, then 4 x 4 steps unrolled, so that the
	// three-stage pipeline's hand-over (f0 <- f1 <- fetch, q0 <- probe) is register renaming instead of a
	// dozen v_mov per step, and the refill test and its merge copies leave the steps.
	uint32_t S = 0;
	clk.mark(0);                                 // (diagnostic build only) prologue done
	// priming steps (latency mode, HD_LAT_PRIME): the front of the pipeline only -- the positions enter the table, the ring
	// fills -- and no tokens: the twin drops them, and no match crosses the border.  (A loop of its own, ahead of the main
	// one: inside it the same lines cost the level-1 kernel 0.8 % of its 16 GiB rate.)
	for (; S < prime && use_static; S += 64) {
		if (filled < n && filled < S + HD_LOOKAHEAD)
			fill_piece();
		f0 = f1;
		q0 = probe(f1.c, S + 64 + lane, f1.c2);
		f1 = fetch(std::false_type{}, S + 128);
		if (OWN_AHEAD)
			o1 = own(S + 192);
	}
	while (S < n && use_static) {
		if (small && filled < n && filled < S + HD_LOOKAHEAD && S + 15 * 64 + 192 + 8 <= n) {
			fill_piece();
#pragma unroll 1
			for (uint32_t g = 0; g < 4 && use_static; g++) {
#pragma unroll
				for (uint32_t u = 0; u < 4; u++) {
					step(std::true_type{}, std::true_type{}, S);
					step_boundary(std::true_type{}, S);
					S += 64;
				}
			}
			continue;
		}
		const bool ok_step = (small && S + 192 + 8 <= n) ? step(std::true_type{}, std::false_type{}, S)
								  : step(std::false_type{}, std::false_type{}, S);
		if (!ok_step || !step_boundary(std::false_type{}, S))
			break;
		S += 64;
	}
	if (use_static && qtail != qhead)
		emit_tokens(qtail - qhead);
	clk.mark(1);                                 // steps done

	// the CRC needs every piece, also when the static stream was abandoned
	while (filled < n) {
		const uint32_t piece = filled / HD_PIECE;
		const uint4 pv = pre;
		filled += HD_PIECE;
		if (filled < n)
			pre = load_slot(src, n, piece + 1, lane, aligned);
		crc.fold(ct, piece, piece * HD_PIECE + lane * 16 + 16 <= n && piece * HD_PIECE + lane * 16 >= prime, pv);
	}
	const uint32_t crcv = crc.finish(ct, lane, n, src + (n & ~15u));
	clk.mark(2);                                 // CRC done

	if (TOK) {
		if (use_static)
			close_deflate_block();
		if (lane == 0) {
			uint32_t *m = (uint32_t *)(rec + lay.off_rec);
			m[0] = use_static ? ndb : 0xffffffffu;       // use_static false here: the slab overflowed
			m[1] = crcv;
			if (!parted)                                 // (a part's slab holds a token per byte)
				a.split_ovf[b] = use_static ? 0u : 1u;
		}
		return;
	}
	if (use_static && (uint64_t)(bitpos - paybase) + 7 > 8ull * limit)
		use_static = false;
	if (!use_static) {
		write_stored_member(a, b, src_own, n_own, crcv, lane);
		return;
	}

	// end-of-block (7 zero bits), pad to a byte, then CRC32 + ISIZE as 4 x 16 bits
	bitpos += 7;
	if (flush) {
		// header of an empty stored block (BFINAL = 0, BTYPE = 00), alignment, LEN = 0, NLEN = ffff
		bitpos = (bitpos + 3 + 7) & ~7u;
		uint32_t tcode = 0, nb = 0;
		if (lane < 2) {
			tcode = lane ? 0xffffu : 0u;
			nb = 16;
		}
		const uint32_t incl = wave_incl_scan<true>(nb);
		put(tcode, nb, incl, 32);
	}
	bitpos = (bitpos + 7) & ~7u;
	const uint32_t paylen = (bitpos - paybase) >> 3;
	if (trl) {
		uint32_t tcode = 0, nb = 0;
		if (lane < trl / 2) {
			tcode = frame_trl_field(a.frame, lane, crcv, n_own);
			nb = 16;
		}
		const uint32_t incl = wave_incl_scan<true>(nb);
		put(tcode, nb, incl, 8 * trl);
	}
	// final flush: everything left, including the last partial dword (< 256 dwords)
	for (uint32_t i = flushed + lane; i < ((bitpos + 31) >> 5); i += 64)
		dst32[i] = stage[i & (STG - 1)];
	if (lane == 0) {
		const uint32_t total = hdr + paylen + trl;
		if (a.frame == HD_FRAME_BGZF)
			*(uint16_t *)((uint8_t *)dst32 + 16) = (uint16_t)(total - 1);
		else if (a.frame == HD_FRAME_MIGZ)
			dst32[4] = paylen;
		a.out_len[b] = total;
		if (a.status) a.status[b] = 0;
		if (a.crc) a.crc[b] = crcv;
	}
}

} // namespace hd
