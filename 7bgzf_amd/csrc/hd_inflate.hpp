// hd_inflate.hpp -- raw-DEFLATE decoder, one wavefront per stream
// (BASELINE config 3: BGZF decode, bit-exact vs the reference).
//
// Replaces, for the hip backend, libdeflate_inflate (lib/zlibutil.c:194-204) ->
// libdeflate_deflate_decompress (lib/libdeflate/decompress_template.h:44-772,
// table build lib/libdeflate/deflate_decompress.c:722-1004), i.e. what
// zlibutil_auto_inflate does per block in applet/7bgzf.c:330.  Accept/reject
// behaviour and result codes follow libdeflate (oracle/hd_inflate.c lists the
// rules with their file:line); the structure does not:
//
//   * tokens are decoded speculatively by the vector units: in a 128-bit window every
//     lane decodes the tokens that would start at its two bit offsets, a short
//     scalar walk (written out in ISA) follows the real chain, literals and most
//     matches are then placed without any per-token scalar work;
//   * around that, a fully checked scalar token loop takes what a window cannot
//     (stream edges, long codewords, end of block): the compressed stream is pulled
//     in 256-byte coalesced pieces (one dword per lane), the bit buffer lives in
//     SGPRs and is refilled with v_readlane;
//   * Huffman tables are built by all 64 lanes (ballot ranks per code length)
//     into LDS: 2^9-entry litlen and 2^8-entry offset tables; longer codewords
//     take a canonical bit-serial slow path instead of subtables;
//   * output goes through an LDS ring; finished 1 KiB pieces leave as 16 B per
//     lane stores; a match copy is 64 bytes per step across the lanes, from the
//     ring when the source is near, from HBM (already flushed) when it is far.
#pragma once
#include <type_traits>
#include "hd_device.hpp"

namespace hd {

struct InflateArgs {
	const uint8_t *in;
	const uint64_t *in_off;
	const uint32_t *in_len;
	uint32_t nblocks;
	uint8_t *out;
	const uint64_t *out_off;
	const uint32_t *out_cap;
	uint32_t *out_len;
	uint32_t *crc;
	int32_t *status;
	const CrcTables *ct;
	uint32_t flags;
};

// InflateArgs::flags: a stream may stop after a non-final block once every input byte is used -- the chunks of
// 7dictzip / 7razf end in a full-flush marker, not in a final block, and the reference reads them with inflaters
// that report "out of input" as success (zlib_inflate, lib/zlibutil.c:289-291; igzip_inflate, zlibutil_igzip.c:111)
constexpr uint32_t INF_FLUSHED = 1;

#ifdef HD_INFLATE_STATS
// experiment build only (tools/exp_inflate_stats.sh): where the tokens go
__device__ unsigned long long g_inf_stats[8];
__device__ unsigned long long g_inf_stats2[8];      // matches of the windows: [0] all [1] lane groups of 8 [2] of 16 [3] simple, one at a time [4] far, one at a time [5] general (fed by the window, overlapping, > 64 bytes)
#define INF_STAT2(k, v) atomicAdd(&g_inf_stats2[k], lane == 0 ? (unsigned long long)(v) : 0ull)
#define INF_U64(m) (((uint64_t)uniform((uint32_t)((m) >> 32)) << 32) | uniform((uint32_t)(m)))
__device__ unsigned long long g_inf_cycles[8];       // [0] block headers + table build, [1] windows, [2] scalar token path, [3] whole kernel
// (every lane adds, lane 0 the value and the others zero: an `if (lane == 0)` around the atomic makes the compiler treat the
// lane masks that live across it as divergent, and the window code keeps them in scalar registers)
#define INF_STAT(k, v) atomicAdd(&g_inf_stats[k], lane == 0 ? (unsigned long long)(v) : 0ull)
#define INF_T0(t) const unsigned long long t = __builtin_amdgcn_s_memtime()
// (cycles are summed in scalar registers and leave once per wave)
#define INF_T1(k, t) do { inf_cyc[k] += __builtin_amdgcn_s_memtime() - (t); } while (0)
#define INF_CYC_DECL unsigned long long inf_cyc[4] = { 0, 0, 0, 0 }
#define INF_CYC_FLUSH do { for (int k_ = 0; k_ < 4; k_++) atomicAdd(&g_inf_cycles[k_], lane == 0 ? inf_cyc[k_] : 0ull); } while (0)
#else
#define INF_STAT(k, v) do { } while (0)
#define INF_STAT2(k, v) do { } while (0)
#define INF_T0(t) do { } while (0)
#define INF_T1(k, t) do { } while (0)
#define INF_CYC_DECL do { } while (0)
#define INF_CYC_FLUSH do { } while (0)
#endif

// experiment switches (tools/exp_inflate_ab.sh rebuilds with -D...): the shipped values are the defaults
#ifndef HD_INF_POLICY
#define HD_INF_POLICY 0      // 1: a 16-lane-group pass only when it pays; 0: always, for every 9..16-byte match
#endif
#ifndef HD_INF_OWNER
#define HD_INF_OWNER 0       // 1: short matches copied by their own lanes (measured: 146 GB/s against 160 with the lane groups --
                             // sixteen ds_write_b8 per window and half cost the LDS path more than the vector units gained); 0: lane groups
#endif
#ifndef HD_INF_SPLIT_SRC
#define HD_INF_SPLIT_SRC 1      // the lane-group passes read "ring byte, or flushed byte" as two typed loads (ds_read_u8 + a rare global_load_ubyte), not one flat load: +1.3 %
#endif
#ifndef HD_INF_ONEPERM
#define HD_INF_ONEPERM 0     // 1: a lane-group pass pushes its owners' words with ONE ds_permute when no lane owns a match in both halves
                             // (round 5, measured: 159.8 / 164.7 / 115.9 GB/s against 161.6 / 166.7 / 117.3 -- the test and the second
                             // code path cost more than the permute and its five vector instructions: profiles/r05_inflate_cuts.txt)
#endif
#ifndef HD_INF_WALK5
#define HD_INF_WALK5 1       // 1: five instructions and one branch per token -- the stop flag is bit 6 of the word that is ADDED to the position, so a
                             // token the walk must stop in front of throws it out of the half by itself and the last token is taken back behind
                             // the loop; 0: rounds 1-4's seven and two.  Round 5, ABAB on one box: 162.4 / 167.5 / 118.4 -> 164.5 / 169.4 / 119.6 GB/s
#endif
#ifndef HD_INF_DEFER
#define HD_INF_DEFER 1       // 1: the first lane-group pass of a window stays open across the scalar copies
#endif
#ifndef HD_INF_PREFETCH
#define HD_INF_PREFETCH 1    // 1: the stream piece after next is requested a refill ahead
#endif

constexpr uint32_t INF_LT_BITS = 9;      // litlen direct table (8 VGPRs once loaded)
constexpr uint32_t INF_DT_BITS = 8;      // offset direct table
constexpr uint32_t INF_RING    = 2048;   // LDS output ring of the throughput kernel: small on purpose, occupancy beats window
                                         // (8 KiB: 36 GB/s, 2 KiB: 49 GB/s on the libdeflate-6 stream)
constexpr uint32_t INF_RING_LAT = 65536; // ... of the latency kernel (k_inflate_lat: a few streams on an empty chip, hip_inflate):
                                         // every match source of a block is in LDS, nothing waits for flushed output

// table entry: [31:16] value (literal / length base / offset base)
//              [9:8] kind  [7:4] extra-bit count  [3:0] codeword length
constexpr uint32_t K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SLOW = 3;

// order in which the precode lengths are stored (RFC 1951 3.2.7;
// decompress_template.h:91-93)
__constant__ uint8_t k_precode_perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len)
{
	if (sym < 256)
		return (sym << 16) | (K_LIT << 8) | len;
	if (sym == 256)
		return (K_EOB << 8) | len;
	// lengths: deflate_decompress.c:563-577 (286, 287 decode as 258)
	uint32_t s = sym - 257, base, eb;
	if (s < 8) { base = 3 + s; eb = 0; }
	else if (s >= 28) { base = 258; eb = 0; }
	else { eb = (s >> 2) - 1; base = 3 + ((4 + (s & 3)) << eb); }
	return (base << 16) | (K_LEN << 8) | (eb << 4) | len;
}

__device__ __forceinline__ uint32_t offset_entry(uint32_t sym, uint32_t len)
{
	// deflate_decompress.c:612-627 (30, 31 decode as 24577 + 13 bits)
	uint32_t base, eb;
	if (sym < 4) { base = 1 + sym; eb = 0; }
	else if (sym >= 30) { base = 24577; eb = 13; }
	else { eb = (sym >> 1) - 1; base = 1 + ((2 + (sym & 1)) << eb); }
	return (base << 16) | (eb << 4) | len;
}

// 6400 bytes = five 1280-byte allocation units = 25 waves per CU (7200 were six units, 21 waves; this
// kernel answers to occupancy: 16 waves measured 85 GB/s where 21 gave 102).  What is live only while a
// block header is read shares its bytes with what is live only in the symbol loop:
//   - the precode's table is the first 128 entries of `lit` (the litlen table is built after the code
//     lengths are complete);
//   - the table builder's per-length scratch lies in cl[320..448), the landing zone of an RLE overrun
//     (decompress_template.h:171), dead once the lengths are in;
//   - cl / pre_lens share the union with the window decoder's dump slots and stream copy; the dump
//     slots follow the ring directly, ring[INF_RING + lane] addresses them.
template <uint32_t RING>
struct InfLdsT {
	uint32_t lit[1u << INF_LT_BITS];
	uint32_t off[1u << INF_DT_BITS];
	uint16_t lit_sorted[288];
	uint16_t off_sorted[32];
	uint16_t lit_count[16], off_count[16];
	union {
		__attribute__((aligned(16))) uint8_t ring[RING];
		uint32_t ring32[RING / 4];
	};
	union {
		struct {
			uint8_t cl[288 + 32 + 138 + 6];   // + worst-case RLE overrun
			uint8_t pre_lens[32];
		};
		struct {
			uint8_t dump[64];                 // a dump slot per lane, right behind the ring
			uint32_t comp[128];               // 2 pieces of the compressed stream for the window decoder
		};
	};
};
using InfLds = InfLdsT<INF_RING>;
static_assert(sizeof(InfLds) == 6400, "InfLds must stay within five LDS allocation units");
constexpr uint32_t INF_T_SCRATCH = 320;      // byte offset in cl of { u32 cnt[16]; u16 first[16]; u16 offs[16]; }

// Build one decode table from code lengths (all 64 lanes).  Returns false for
// what build_decode_table() rejects (deflate_decompress.c:799-853).
//   nsyms <= 288; tbits = direct table bits; kind 0 litlen / 1 offset / 2 precode
// Per-length counters live in LDS (one lane per code length), not in registers:
// sixteen-element uniform arrays would cost 64 SGPRs and spill the symbol loop.
template <int KIND>
__device__ __noinline__ uint32_t build_table(const uint8_t *lens, uint32_t nsyms, uint32_t *table, uint32_t tbits,
					  uint16_t *sorted, uint16_t *count_out, uint8_t *scratch, uint32_t lane)
{
	// per-length counters; once the counts are read they serve as the running rank bases
	uint32_t *t_cnt = (uint32_t *)scratch, *t_base = t_cnt;
	uint16_t *t_first = (uint16_t *)(scratch + 64), *t_offs = t_first + 16;
	if (lane < 16)
		t_cnt[lane] = 0;
	for (uint32_t base = 0; base < nsyms; base += 64) {
		const uint32_t s = base + lane;
		const uint32_t len = s < nsyms ? lens[s] : 0;
		if (len)
			atomicAdd(&t_cnt[len], 1u);
	}
	// lane l (1..15) derives its length's first codeword, sorted offset and codespace share
	uint32_t code = 0, o = 0, mine = 0;
	if (lane >= 1 && lane < 16) {
		for (uint32_t l = 1; l <= lane; l++) {
			code = (code + (l > 1 ? t_cnt[l - 1] : 0)) << 1;
			if (l < lane)
				o += t_cnt[l];
		}
		mine = t_cnt[lane];
		t_first[lane] = (uint16_t)code;       // < 2^15 for every code that passes the checks below
		t_offs[lane] = (uint16_t)o;
	}
	if (lane < 16)
		t_base[lane] = 0;                     // (all lanes are past the loop that read the counts)
	if (count_out && lane < 16)
		count_out[lane] = (uint16_t)mine;
	const uint32_t used = readlane(wave_incl_scan((lane >= 1 && lane < 16) ? mine << (15 - lane) : 0u), 63);
	const uint64_t present = __ballot(mine != 0);
	const uint32_t maxlen = present ? 63 - (uint32_t)__clzll((long long)present) : 0;
	const uint32_t ones = readlane(mine, 1);
	if (used > (1u << 15))
		return 0;                         // overfull
	bool degenerate = false;
	if (used < (1u << 15)) {                  // incomplete
		if (used != 0 && !(ones == 1 && maxlen == 1))
			return 0;
		degenerate = true;
	}
	uint32_t one_sym = 0;                 // the symbol owning the single 1-bit codeword, if any
	for (uint32_t base = 0; base < nsyms; base += 64) {
		const uint32_t s = base + lane;
		const uint32_t len = s < nsyms ? lens[s] : 0;
		// rank inside the length class, in symbol order: one ballot per length present
		uint32_t rank = 0;
		uint64_t rem = __ballot(len != 0);
		while (rem) {
			const uint32_t lead = (uint32_t)__ffsll((unsigned long long)rem) - 1;
			const uint32_t lc = readlane(len, lead);
			const uint64_t m = __ballot(len == lc);
			const uint32_t b0 = uniform(t_base[lc]);
			if (len == lc)
				rank = b0 + __popcll(m & ((1ull << lane) - 1));
			if (lane == lead)
				t_base[lc] = b0 + __popcll(m);
			if (lc == 1)
				one_sym = base + lead;
			rem &= ~m;
		}
		if (len) {
			sorted[t_offs[len] + rank] = (uint16_t)s;
			const uint32_t cw = t_first[len] + rank;        // MSB-first codeword
			const uint32_t rev = __brev(cw) >> (32 - len);
			if (!degenerate) {
				if (len <= tbits) {
					const uint32_t e = KIND == 0 ? litlen_entry(s, len)
							 : KIND == 1 ? offset_entry(s, len)
								     : ((s << 16) | len);
					for (uint32_t i = rev; i < (1u << tbits); i += 1u << len)
						table[i] = e;
				} else {
					table[rev & ((1u << tbits) - 1)] = K_SLOW << 8;
				}
			}
		}
	}
	if (degenerate) {
		// empty code -> symbol 0; single 1-bit codeword -> that symbol, for both
		// bit values (deflate_decompress.c:816-849)
		const uint32_t sym = used ? one_sym : 0;
		const uint32_t e = KIND == 0 ? litlen_entry(sym, 1) : KIND == 1 ? offset_entry(sym, 1) : ((sym << 16) | 1);
		for (uint32_t i = lane; i < (1u << tbits); i += 64)
			table[i] = e;
	}
	return 1;
}

// Canonical decode for codewords longer than the direct table; returns symbol | length << 16.
// All fifteen lengths are tried at once: lane l holds count[l], two prefix sums give its length's first
// codeword (first[l] = sum_{k<l} count[k] << (l - k), i.e. the Kraft sum in 2^-15 units shifted back) and
// its offset into the sorted symbols, one ballot finds the length that matches.  (The bit-serial loop it
// replaces paid an LDS round trip per length: ~1 M such tokens per GiB of a libdeflate-6 stream.)
__device__ __noinline__ uint32_t slow_decode(uint64_t bb, const uint16_t *count, const uint16_t *sorted, uint32_t lane)
{
	const bool in = lane >= 1 && lane < 16;
	const uint32_t l = lane & 15;
	const uint32_t cnt = in ? count[l] : 0u;
	const uint32_t scaled = cnt << (15 - l);
	const uint32_t first = (wave_incl_scan(scaled) - scaled) >> (15 - l);
	const uint32_t offs = wave_incl_scan(cnt) - cnt;
	const uint32_t code = __brev((uint32_t)bb) >> (32 - (l ? l : 1));     // the first l stream bits, MSB first
	const uint32_t d = code - first;
	const uint64_t hit = __ballot(in && d < cnt);
	if (!hit)
		return 15u << 16;
	const uint32_t len = (uint32_t)__ffsll((unsigned long long)hit) - 1;
	return (uint32_t)sorted[readlane(offs + d, len)] | (len << 16);
}

// the decoder, for an output ring of RING bytes (one wavefront; L is the workgroup's LDS)
template <uint32_t RING>
__device__ __forceinline__ void inflate_stream(const InflateArgs &a, InfLdsT<RING> &L)
{
	constexpr uint32_t INF_NEAR = RING - 258 - 64;       // dist <= this: source is in the ring
	const uint32_t lane = threadIdx.x;
	const uint32_t b = blockIdx.x;
	if (b >= a.nblocks)
		return;
	const ClockStamp clk(HD_CLK_INFLATE);
	const uint8_t *src = a.in + a.in_off[b];
	const uint32_t n = a.in_len[b];
	if (n >= HD_INFLATE_MAX_IN) {
		// stream positions are 32-bit BIT counts (over_t, B below): a stream this long is refused whole
		// rather than decoded from wrapped positions (the host entry points answer HD_E_ARG before launching)
		if (lane == 0) {
			a.out_len[b] = 0;
			if (a.crc) a.crc[b] = 0;
			if (a.status) a.status[b] = HD_BAD_DATA;
		}
		return;
	}
	uint8_t *dst = a.out + a.out_off[b];
	const uint32_t cap = a.out_cap[b];
	const CrcTables *ct = a.ct;
	const bool want_crc = a.crc != nullptr;
	const bool dst_aligned = (((uintptr_t)dst) & 15) == 0;
	[[maybe_unused]] const bool dst_al4 = (((uintptr_t)dst) & 3) == 0;
	[[maybe_unused]] const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)L.ring;   // LDS byte address of the ring

	// ---- compressed input: 256-byte pieces, one dword per lane -----------
	const uint32_t mis = (uint32_t)((uintptr_t)src & 3);
	const uint32_t *src32 = (const uint32_t *)(src - mis);
	const uint32_t nbytes_al = mis + n;                  // valid bytes from src32
	auto load_piece = [&](uint32_t piece) -> uint32_t {
		const uint32_t d = piece * 64 + lane;
		uint32_t w = 0;
		if (d * 4 < nbytes_al) {
			w = src32[d];
			const uint32_t valid = nbytes_al - d * 4;    // bytes of this dword inside the stream
			if (valid < 4)
				w &= (1u << (8 * valid)) - 1;
		}
		return w;
	};
	uint32_t cur_piece = 0;
	uint32_t cw = load_piece(0), cw_next = load_piece(1);
	uint32_t dw = 0;                 // next dword index to feed the bit buffer
	uint64_t bb = 0;                 // bit buffer (uniform)
	uint32_t bc = 0;                 // valid bits in bb

	auto next_dword = [&]() -> uint32_t {
		const uint32_t piece = dw >> 6;
		if (piece != cur_piece) {        // uniform branch: step to the next piece
			cw = cw_next;
			cur_piece = piece;
			cw_next = load_piece(piece + 1);
		}
		const uint32_t w = readlane(cw, dw & 63);
		dw++;
		return w;
	};
	auto refill = [&]() {
		if (bc <= 32) {
			bb |= (uint64_t)next_dword() << bc;
			bc += 32;
		}
	};
	auto consumed_bits = [&]() -> int64_t { return (int64_t)dw * 32 - bc - 8 * (int64_t)mis; };
	// consumed_bits() > 8 n + 64 in 32-bit arithmetic (n < 2^28): (dw << 5) - bc > over_t
	const uint32_t over_t = 8 * n + 64 + 8 * mis;
	auto overrun = [&]() -> bool { return (dw << 5) - bc > over_t; };
	auto seek_byte = [&](uint32_t byteoff) {       // restart the bit reader at src + byteoff
		const uint32_t o = mis + byteoff;
		dw = o >> 2;
		const uint32_t piece = dw >> 6;
		if (piece != cur_piece) {
			cur_piece = piece;
			cw = load_piece(piece);
			cw_next = load_piece(piece + 1);
		}
		bb = 0;
		bc = 0;
		refill();
		bb >>= 8 * (o & 3);
		bc -= 8 * (o & 3);
	};
	// skip the mis-alignment bytes
	refill();
	bb >>= 8 * mis;
	bc -= 8 * mis;

	// ---- output ring + flush ---------------------------------------------
	uint32_t pos = 0, flushed = 0;
	CrcLanes crc;
	crc.init(lane, 0xffffffffu);          // length unknown yet; lane 0 seeds, fixed in finish
	auto flush_pieces = [&]() {
		while (pos - flushed >= HD_PIECE) {
			const uint4 v = *(const uint4 *)&L.ring[(flushed & (RING - 1)) + 16 * lane];
			if (dst_aligned) {
				*(uint4 *)(dst + flushed + 16 * lane) = v;
			} else {
				const uint32_t w[4] = { v.x, v.y, v.z, v.w };
				for (uint32_t k = 0; k < 16; k++)
					dst[flushed + 16 * lane + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
			}
			if (want_crc)
				crc.fold(ct, flushed / HD_PIECE, true, v);
			flushed += HD_PIECE;
		}
	};


	// ---- window decode: 64 speculative tokens per pass -----------------------
	// The scalar token loop costs ~80 SALU per token on the CU's single scalar ALU.
	// Here every lane decodes the token that WOULD start at bit B + lane (two LDS
	// table gathers, VALU only); a short scalar walk then follows the real chain
	// (~10 SALU per token), literals are stored in parallel at positions from a
	// DPP prefix sum, matches are copied in order.  A window is entered only
	// when nothing rare can happen inside it: five whole dwords of stream ahead,
	// some room in the output (the window's budget), less than a piece waiting for the flush.
	// Returns 0 = fall back to the scalar loop for one token, 1 = end of block
	// consumed, 2 = error (st set).  Reader state is the scalar one on both sides.
	// output budget of one window: pending <= 1023 + 704 = 1727 <= RING - 64 - 257, so a source is
	// either wholly in the ring (wend - src <= RING - 64) or wholly flushed to HBM
	constexpr uint32_t WIN_OUT_BUDGET = 704;
	const uint32_t dw_safe = (mis + n) >> 2;      // dwords below this are whole
	uint32_t lds_p0 = 0xfffffff0u;                // pieces lds_p0, lds_p0 + 1 are in L.comp
	uint32_t pre_piece = 0, pre_idx = 0xfffffff0u; // piece pre_idx of the stream, requested ahead of its use
	auto run_windows = [&](int32_t &st_out) -> uint32_t {
		uint32_t B = (dw << 5) - bc;              // absolute bit position from src32
		uint32_t result = 0;
		for (;;) {
			if (pos - flushed >= HD_PIECE)
				flush_pieces();
			const uint32_t d0 = B >> 5;
			// the budget shrinks to the room that is left, so windows run up to the last bytes of the
			// output (a token that does not fit is cut below and meets the scalar loop's checks)
			const uint32_t budget = cap - pos < WIN_OUT_BUDGET ? cap - pos : WIN_OUT_BUDGET;
			if (!(d0 + 7 <= dw_safe && budget != 0))
				break;
			// the stream bits come from an LDS copy of the pieces around d0 (every lane
			// reads its own dwords: no scalar gather)
			// (piece p0 always sits in comp[0,64) and p0 + 1 in comp[64,128): the five dwords under a lane's two
			// decodes are consecutive and never wrap -- one address, three LDS reads for both)
			const uint32_t p0 = d0 >> 6;
			if (p0 != lds_p0) {
				if (p0 == lds_p0 + 1)
					L.comp[lane] = L.comp[64 + lane];
				else
					L.comp[lane] = load_piece(p0);
				// the piece behind was requested when the last one was put in: its load has had ~16 windows to arrive
				if (HD_INF_PREFETCH) {
					L.comp[64 + lane] = pre_idx == p0 + 1 ? pre_piece : load_piece(p0 + 1);
					pre_piece = load_piece(p0 + 2);
					pre_idx = p0 + 2;
				} else {
					L.comp[64 + lane] = load_piece(p0 + 1);
				}
				lds_p0 = p0;
			}
			// A window is 128 bits: every lane decodes the token that would start at bit
			// B + lane ("lo") and the one at B + 64 + lane ("hi").  Twice the tokens per
			// window halve the scalar glue per token, which is what bounds this kernel.
			struct Spec {
				uint32_t e, length, offset, outlen, walk;
				uint64_t is_len, is_lit;             // lane masks (one v_cmp each, used through sel())
			};
			// dwords w[0..4] under bit (B & 31) + lane: the "lo" decode reads w[0..2], the "hi" one (64 bits on) w[2..4]
			const uint32_t bl0 = (B & 31) + lane;
			const uint32_t *wsp = &L.comp[(d0 & 63) + (bl0 >> 5)];
			const uint32_t ws0 = wsp[0], ws1 = wsp[1], ws2 = wsp[2], ws3 = wsp[3], ws4 = wsp[4];
			auto spec = [&](uint32_t bl, uint32_t lo, uint32_t mid, uint32_t hi) -> Spec {   // bl = bit offset from dword d0
				Spec r;
				const uint32_t a = __builtin_amdgcn_alignbit(mid, lo, bl & 31);
				const uint32_t bq = __builtin_amdgcn_alignbit(hi, mid, bl & 31);
				const uint32_t e = L.lit[a & ((1u << INF_LT_BITS) - 1)];
				const uint32_t len1 = e & 15, eb = (e >> 4) & 15;
				r.e = e;
				r.length = (e >> 16) + __builtin_amdgcn_ubfe(a, len1, eb);         // (one v_bfe_u32; width 0 gives 0)
				const uint32_t t1 = len1 + eb;                 // <= 9 + 5
				const uint32_t rest = __builtin_amdgcn_alignbit(bq, a, t1);
				const uint32_t dd = L.off[rest & ((1u << INF_DT_BITS) - 1)];
				const uint32_t kind2 = e & 0x300;
				r.is_len = __ballot(kind2 == (K_LEN << 8));
				r.is_lit = __ballot(kind2 == (K_LIT << 8));
				// (the offset entry counts for a length only: masked here, its fields are zero elsewhere -- and so is
				// eb in a literal's or an end-of-block's entry, so the token's bits are one sum)
				const uint32_t ddm = sel(r.is_len, dd, 0u);
				const uint32_t dlen = ddm & 15, deb = (ddm >> 4) & 15;
				r.offset = (ddm >> 16) + __builtin_amdgcn_ubfe(rest, dlen, deb);
				const uint32_t tokbits = t1 + dlen + deb;
				r.outlen = sel(r.is_lit, 1u, sel(r.is_len, r.length, 0u));
				// bit 6 = the walk stops in front of this token: bit 9 of either entry (K_EOB and K_SLOW have it)
				// moved down.  (A zero-bit token cannot come out of a well-formed table; the max keeps the walk
				// moving whatever the table holds.)
				const uint32_t tb1 = tokbits ? tokbits : 1u;
				r.walk = tb1 | ((e >> 3) & 64u) | ((ddm >> 3) & 64u);
				return r;
			};
			const Spec s0 = spec(bl0, ws0, ws1, ws2), s1 = spec(bl0 + 64, ws2, ws3, ws4);

			// The real chain from bit 0 of the window.  This walk is the hottest scalar
			// code of the kernel (the CU has one scalar ALU) and the compiler spends ~20
			// instructions per token on it, so it is written out: 4 SALU + 2 branches
			// + 1 v_readlane per token (the lane select of v_readlane and s_bitset1 take
			// the low 6 bits, so the second half runs on b itself).  It stops in front of
			// the first token the window cannot take; that one goes to the scalar loop.
			// (A lane select written by the SALU needs no wait states before v_readlane,
			// only one written by the VALU does.)
			uint32_t b, wm;
			uint64_t real0, real1;
#if HD_INF_WALK5
			asm volatile("s_mov_b32 %0, 0\n\t"
				     "s_mov_b64 %1, 0\n\t"
				     "s_mov_b64 %2, 0\n"
				     "Lhd_walk0_%=:\n\t"
				     "v_readlane_b32 %3, %4, %0\n\t"
				     "s_bitset1_b64 %1, %0\n\t"
				     "s_add_u32 %0, %0, %3\n\t"
				     "s_cmp_lt_u32 %0, 64\n\t"
				     "s_cbranch_scc1 Lhd_walk0_%=\n\t"
				     "s_bitcmp1_b32 %3, 6\n\t"
				     "s_cbranch_scc0 Lhd_walk1_%=\n\t"
				     "s_sub_u32 %0, %0, %3\n\t"
				     "s_bitset0_b64 %1, %0\n\t"
				     "s_branch Lhd_walk_done_%=\n"
				     "Lhd_walk1_%=:\n\t"
				     "v_readlane_b32 %3, %5, %0\n\t"
				     "s_bitset1_b64 %2, %0\n\t"
				     "s_add_u32 %0, %0, %3\n\t"
				     "s_cmp_lt_u32 %0, 128\n\t"
				     "s_cbranch_scc1 Lhd_walk1_%=\n\t"
				     "s_bitcmp1_b32 %3, 6\n\t"
				     "s_cbranch_scc0 Lhd_walk_done_%=\n\t"
				     "s_sub_u32 %0, %0, %3\n\t"
				     "s_bitset0_b64 %2, %0\n"
				     "Lhd_walk_done_%=:"
				     : "=&s"(b), "=&s"(real0), "=&s"(real1), "=&s"(wm)
				     : "v"(s0.walk), "v"(s1.walk)
				     : "scc");
#else
			asm volatile("s_mov_b32 %0, 0\n\t"
				     "s_mov_b64 %1, 0\n\t"
				     "s_mov_b64 %2, 0\n"
				     "Lhd_walk0_%=:\n\t"
				     "v_readlane_b32 %3, %4, %0\n\t"
				     "s_bitcmp1_b32 %3, 6\n\t"
				     "s_cbranch_scc1 Lhd_walk_done_%=\n\t"
				     "s_bitset1_b64 %1, %0\n\t"
				     "s_add_u32 %0, %0, %3\n\t"
				     "s_cmp_lt_u32 %0, 64\n\t"
				     "s_cbranch_scc1 Lhd_walk0_%=\n"
				     "Lhd_walk1_%=:\n\t"
				     "v_readlane_b32 %3, %5, %0\n\t"
				     "s_bitcmp1_b32 %3, 6\n\t"
				     "s_cbranch_scc1 Lhd_walk_done_%=\n\t"
				     "s_bitset1_b64 %2, %0\n\t"
				     "s_add_u32 %0, %0, %3\n\t"
				     "s_cmp_lt_u32 %0, 128\n\t"
				     "s_cbranch_scc1 Lhd_walk1_%=\n"
				     "Lhd_walk_done_%=:"
				     : "=&s"(b), "=&s"(real0), "=&s"(real1), "=&s"(wm)
				     : "v"(s0.walk), "v"(s1.walk)
				     : "scc");
#endif
			// output positions; cut in front of the first token that would overrun the budget.  (Conditions are
			// 64-bit lane masks: the ballot of ONE compare each, combined in scalar code, back to the lanes
			// through sel() -- hd_device.hpp "lane masks".)
			// (both halves' output lengths in one prefix sum, 16 bits each: 64 x 258 < 2^16)
			const uint32_t scn = wave_incl_scan(sel(real0, s0.outlen, 0u) | (sel(real1, s1.outlen, 0u) << 16));
			const uint32_t tot = readlane(scn, 63);                   // the totals of both halves
			const uint32_t incl0 = scn & 0xffff;
			const uint32_t incl1 = (scn >> 16) + (tot & 0xffff);
			// (the usual window fits its budget whole: one scalar compare, and the total is its output)
			uint32_t cum = (tot & 0xffff) + (tot >> 16);
			const bool over = cum > budget;
			if (over) {
				const uint64_t over0 = __ballot(incl0 > budget) & real0;
				const uint64_t over1 = __ballot(incl1 > budget) & real1;
				if (over0) {
					b = (uint32_t)__ffsll((unsigned long long)over0) - 1;
					real0 &= (1ull << b) - 1;
					real1 = 0;
				} else {
					const uint32_t f = (uint32_t)__ffsll((unsigned long long)over1) - 1;
					real1 &= (1ull << f) - 1;
					b = 64 + f;
				}
				if (real0)
					cum = real1 ? readlane(incl1, 63 - (uint32_t)__clzll((long long)real1))
						    : readlane(incl0, 63 - (uint32_t)__clzll((long long)real0));
			}
			INF_STAT(0, 1);
			if (real0 == 0) {
				INF_STAT(6, 1);
				break;                                     // the token at B is not for a window: scalar loop
			}
			INF_STAT(1, __popcll(real0) + __popcll(real1));
			const uint32_t rel0 = incl0 - s0.outlen, rel1 = incl1 - s1.outlen;   // valid on the real tokens
			const uint32_t opos0 = pos + rel0, opos1 = pos + rel1;
			// (lanes without a literal write to their dump slot: no exec juggling, no skip branches)
			const uint64_t lit0 = real0 & s0.is_lit, lit1 = real1 & s1.is_lit;
			auto store_literals = [&]() {
				L.ring[sel(lit0, opos0 & (RING - 1), RING + lane)] = (uint8_t)(s0.e >> 16);
				L.ring[sel(lit1, opos1 & (RING - 1), RING + lane)] = (uint8_t)(s1.e >> 16);
			};
			if (!HD_INF_OWNER)
				store_literals();
			const uint32_t wend = pos + cum;
			// per-lane verdicts for the matches, so that the scalar loops below only dispatch
			const uint64_t match0 = real0 & s0.is_len, match1 = real1 & s1.is_len;
			const uint32_t srcl0 = opos0 - s0.offset, srcl1 = opos1 - s1.offset;   // wrap when offset > opos
			// (one v_cmp each: written out, or the compiler takes the carry of the subtraction above and turns it
			// back into a mask with two more instructions)
			uint64_t far0, far1;
			asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(far0) : "v"(opos0), "v"(s0.offset));
			asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(far1) : "v"(opos1), "v"(s1.offset));
			if ((far0 & match0) | (far1 & match1)) {   // offset > bytes out so far: decompress_template.h:724
				st_out = HD_BAD_DATA;
				result = 2;
				break;
			}
			// "simple": source wholly in the ring and wholly in front of this window's output,
			// at most 64 bytes.  Nothing in the window feeds them, so they go first and in any order;
			// whatever else there is follows in stream order.
			const uint64_t inr0 = __ballot(wend - srcl0 <= RING - 64), inr1 = __ballot(wend - srcl1 <= RING - 64);
			const uint64_t le0 = __ballot(s0.length <= 64), le1 = __ballot(s1.length <= 64);
			const uint64_t simple0 = match0 & inr0 & __ballot(s0.offset >= rel0 + s0.length) & le0;
			const uint64_t simple1 = match1 & inr1 & __ballot(s1.offset >= rel1 + s1.length) & le1;
			// "far": the source left the ring long ago (it ends >= 1217 bytes in front of this window and is flushed:
			// pending < 1024, budget <= 704) -- nothing in the window feeds it either, but a load from HBM takes ~700 ns.
			// A libdeflate-6 stream of FASTQ-like data has several per window (5 vector loads per 64 output bytes,
			// 54 % of a wave's cycles parked on s_waitcnt): their loads all go out HERE, up to four at a time, and
			// the bytes are put into the ring behind the ring-to-ring copies below instead of one round trip each.
			const uint64_t hbm0 = match0 & ~inr0 & le0, hbm1 = match1 & ~inr1 & le1;
			// LANE GROUPS.  Most matches of a DEFLATE stream are short (libdeflate-6 on FASTQ-like data: 70 % <= 8 bytes,
			// 90 % <= 16), and both kinds above -- "simple" and "far" -- depend on nothing in this window.  They are copied
			// several at a time, a group of 8 (16) lanes per match, with no scalar work per match: every owner pushes
			// one dword {offset in the window's output, length, distance} to the first lane of its group (ds_permute),
			// the group fetches it (ds_bpermute), each lane moves one byte -- from the ring, or, for a source that has
			// left the ring, straight from the flushed output in HBM (one vector load for up to eight matches, where
			// the scalar path below needs a load, three v_readlane and a dozen scalar instructions per match).
			const uint64_t l8_0 = __ballot(s0.length <= 8), l8_1 = __ballot(s1.length <= 8);
			const uint64_t l16_0 = __ballot(s0.length <= 16), l16_1 = __ballot(s1.length <= 16);
#if HD_INF_OWNER
			// OWNER COPIES.  A short match (<= 16 bytes) of either kind is copied by ITS OWN LANE: three (five) aligned
			// dwords from the source -- the ring, or the flushed output in HBM --, v_alignbyte, then the bytes one
			// ds_write_b8 each with the byte number as the instruction's offset, for all such matches of the window
			// at once.  All 8 (16) bytes are written whatever the length, highest byte first: what a match writes
			// beyond its length lands on the bytes of the tokens behind it, and those are written LATER -- the lower
			// bytes of the other owners by the instructions that follow, the literals and every other match after
			// this block -- or, beyond the window's end, in the ring's 64 oldest bytes, which no source may touch.
			// No compaction, no LDS permutes, no per-match scalar work: ~25 vector instructions for all of them where
			// the lane groups took ~36 per pass (INSTS_VALU 209 -> 199 per 64 bytes with the groups; this kernel is
			// bound by vector issue).  Not for: a destination run that wraps the ring (one scalar test per window),
			// a ring source whose five dwords would, a far source when the output is not 4-byte aligned.
			const bool ring_room = (pos & (RING - 1)) + cum + 16 <= RING;
			const uint64_t sw0 = __ballot((srcl0 & (RING - 1)) <= RING - 20), sw1 = __ballot((srcl1 & (RING - 1)) <= RING - 20);
			const uint64_t vfar0 = dst_al4 ? hbm0 : 0ull, vfar1 = dst_al4 ? hbm1 : 0ull;
			const uint64_t vec0 = ring_room ? ((simple0 & sw0) | vfar0) & l16_0 : 0ull;
			const uint64_t vec1 = ring_room ? ((simple1 & sw1) | vfar1) & l16_1 : 0ull;
#else
			const uint64_t vec0 = (simple0 | hbm0) & l16_0, vec1 = (simple1 | hbm1) & l16_1;
#endif
			uint64_t fa = hbm0 & ~vec0, fb = hbm1 & ~vec1;            // far and not taken above: the scalar path
			// (FARK in flight; every one holds a register, and at 81 the kernel would lose a wave per SIMD -- the offset
			// table's copy in registers made room: the scalar loop reads it from LDS now)
			constexpr int FARK = 1;
			uint32_t fml[FARK], fP[FARK], fv[FARK];
			auto far_issue = [&]() {
#pragma unroll
				for (int k = 0; k < FARK; k++) {
					uint32_t sp = 0;
					fml[k] = 0;
					fP[k] = 0;
					if (fa) {
						const uint32_t m = (uint32_t)__ffsll((unsigned long long)fa) - 1;
						asm("s_bitset0_b64 %0, %1" : "+s"(fa) : "s"(m));
						fml[k] = readlane(s0.outlen, m);
						fP[k] = readlane(opos0, m);
						sp = readlane(srcl0, m);
					} else if (fb) {
						const uint32_t m = (uint32_t)__ffsll((unsigned long long)fb) - 1;
						asm("s_bitset0_b64 %0, %1" : "+s"(fb) : "s"(m));
						fml[k] = readlane(s1.outlen, m);
						fP[k] = readlane(opos1, m);
						sp = readlane(srcl1, m);
					}
					fv[k] = lane < fml[k] ? (uint32_t)dst[sp + lane] : 0u;
				}
			};
			auto far_store = [&]() {
#pragma unroll
				for (int k = 0; k < FARK; k++)
					L.ring[lane < fml[k] ? ((fP[k] + lane) & (RING - 1)) : RING + lane] = (uint8_t)fv[k];
			};
			const bool anyfar = (hbm0 | hbm1) != 0;
			if (anyfar)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // our own stores first
			const bool scalar_far = (fa | fb) != 0;
			if (scalar_far)
				far_issue();
			uint64_t done0 = 0, done1 = 0;                           // matches the lane groups took
			bool pend = false;                                       // a lane-group pass is open: bytes pend_v for ring[pend_idx]
			uint32_t pend_idx = 0, pend_v = 0;
#if HD_INF_OWNER
			{
				auto owner_copy = [&](uint64_t vecm, uint64_t longm, uint64_t farm, uint32_t oposv, uint32_t srclv) {
					if ((vecm >> lane) & 1) {
						const uint32_t al = srclv & ~3u, sh = srclv & 3u;
						uint32_t w0, w1, w2, w3, w4;
						if ((farm >> lane) & 1) {
							const uint32_t *g = (const uint32_t *)(dst + al);
							w0 = g[0]; w1 = g[1]; w2 = g[2]; w3 = g[3]; w4 = g[4];
						} else {
							const uint32_t ri = (al & (RING - 1)) >> 2;
							w0 = L.ring32[ri]; w1 = L.ring32[ri + 1]; w2 = L.ring32[ri + 2]; w3 = L.ring32[ri + 3]; w4 = L.ring32[ri + 4];
						}
						const uint32_t da = ring_lds + (oposv & (RING - 1));
						if ((longm >> lane) & 1) {
							const uint32_t b2 = __builtin_amdgcn_alignbyte(w3, w2, sh), b3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
							asm volatile("ds_write_b8_d16_hi %0, %4 offset:15\n\t"
								     "ds_write_b8_d16_hi %0, %3 offset:14\n\t"
								     "ds_write_b8 %0, %4 offset:13\n\t"
								     "ds_write_b8 %0, %3 offset:12\n\t"
								     "ds_write_b8_d16_hi %0, %2 offset:11\n\t"
								     "ds_write_b8_d16_hi %0, %1 offset:10\n\t"
								     "ds_write_b8 %0, %2 offset:9\n\t"
								     "ds_write_b8 %0, %1 offset:8"
								     :: "v"(da), "v"(b2), "v"(b2 >> 8), "v"(b3), "v"(b3 >> 8) : "memory");
						}
						const uint32_t b0 = __builtin_amdgcn_alignbyte(w1, w0, sh), b1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
						asm volatile("ds_write_b8_d16_hi %0, %4 offset:7\n\t"
							     "ds_write_b8_d16_hi %0, %3 offset:6\n\t"
							     "ds_write_b8 %0, %4 offset:5\n\t"
							     "ds_write_b8 %0, %3 offset:4\n\t"
							     "ds_write_b8_d16_hi %0, %2 offset:3\n\t"
							     "ds_write_b8_d16_hi %0, %1 offset:2\n\t"
							     "ds_write_b8 %0, %2 offset:1\n\t"
							     "ds_write_b8 %0, %1"
							     :: "v"(da), "v"(b0), "v"(b0 >> 8), "v"(b1), "v"(b1 >> 8) : "memory");
					}
				};
				if (vec0)
					owner_copy(vec0, vec0 & ~l8_0, hbm0, opos0, srcl0);
				if (vec1)
					owner_copy(vec1, vec1 & ~l8_1, hbm1, opos1, srcl1);
				store_literals();
				done0 = vec0;
				done1 = vec1;
			}
#else
			if (vec0 | vec1) {
				// rel < 1024 (the window's budget), length <= 16, distance <= 32768: 10 + 5 + 16 bits
				const uint32_t pk0 = rel0 | (s0.length << 10) | (s0.offset << 15);
				const uint32_t pk1 = rel1 | (s1.length << 10) | (s1.offset << 15);
				// The FIRST pass of a window is left open: its reads -- ring bytes, and for far sources one vector load from
				// the flushed output -- are issued here, the bytes go into the ring behind the scalar copies below, so
				// the load's ~1 us is not waited for where it is issued.  (One open pass = two registers; the kernel sits
				// three below the step that costs a wave per SIMD.)
				auto group_pass = [&](auto gtag, uint64_t own0, uint64_t own1) {
					constexpr uint32_t G = decltype(gtag)::value, NG = 64 / G;
					const uint32_t n0 = (uint32_t)__popcll(own0), nt = n0 + (uint32_t)__popcll(own1);
					const uint32_t slot0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(own0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)own0, 0));
					const uint32_t slot1 = n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(own1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)own1, 0));
					const uint32_t sub = lane & (G - 1), lead = (lane & ~(G - 1)) << 2;
					const bool both = (own0 & own1) != 0;
					for (uint32_t base = 0; base < nt; base += NG) {
						// an owner whose slot falls into this pass targets the first lane of group (slot - base); everybody
						// else an odd lane (never a group's first): what arrives there is not looked at
						uint32_t gg;
						if (HD_INF_ONEPERM && !both) {
							// (no lane owns a match in both halves -- nine windows in ten: one push instead of two)
							const uint32_t d = sel(own0, slot0 - base, sel(own1, slot1 - base, NG));
							const uint32_t ad = d < NG ? d * (4 * G) : ((lane | 1u) << 2);
							gg = (uint32_t)__builtin_amdgcn_ds_permute((int)ad, (int)(d < NG ? sel(own0, pk0, pk1) : 0u));
						} else {
							const uint32_t d0 = sel(own0, slot0 - base, NG), d1 = sel(own1, slot1 - base, NG);
							const uint32_t a0 = d0 < NG ? d0 * (4 * G) : ((lane | 1u) << 2), a1 = d1 < NG ? d1 * (4 * G) : ((lane | 1u) << 2);
							const uint32_t g0 = (uint32_t)__builtin_amdgcn_ds_permute((int)a0, (int)(d0 < NG ? pk0 : 0u));
							const uint32_t g1 = (uint32_t)__builtin_amdgcn_ds_permute((int)a1, (int)(d1 < NG ? pk1 : 0u));
							gg = g0 | g1;
						}
						const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)lead, (int)gg);
						const uint32_t ml = (w >> 10) & 31;                        // 0: no match in this group
						const uint32_t dp = pos + (w & 1023) + sub, sp = dp - (w >> 15);
						const bool act = sub < ml;
						// (the whole source in the ring or the whole source flushed: the same test as inr above)
						const bool ringsrc = wend - (sp - sub) <= RING - 64;
#if HD_INF_SPLIT_SRC
						// (said with the address spaces: left generic, the compiler turns "ring byte, or the flushed byte where the source is far"
						// into ONE flat_load_ubyte from a selected address -- every pass then goes through the flat path, LDS and memory counters both)
						uint32_t v = *((const __attribute__((address_space(3))) uint8_t *)&L.ring[0] + (sp & (RING - 1)));
						if (act && !ringsrc)
							v = *((const __attribute__((address_space(1))) uint8_t *)dst + sp);
#else
						uint32_t v = L.ring[sp & (RING - 1)];
						if (act && !ringsrc)
							v = dst[sp];
#endif
						const uint32_t di = act ? (dp & (RING - 1)) : RING + lane;
						if (HD_INF_DEFER && !pend) {
							pend = true;
							pend_idx = di;
							pend_v = v;
						} else {
							L.ring[di] = (uint8_t)v;
						}
					}
				};
				// A pass costs the same for one match as for a full set of groups (three LDS permutes, ~35 vector
				// instructions): eight-lane groups take the <= 8-byte matches; the 9..16-byte ones get a pass of
				// sixteen-lane groups when it pays -- a far one among them (the scalar far path is a load and ~45
				// instructions per match), three or more, or when everything fits one such pass (<= 4 matches);
				// otherwise they are "simple" and go one at a time below.
				uint64_t a0 = vec0 & l8_0, a1 = vec1 & l8_1, b0 = vec0 & ~l8_0, b1 = vec1 & ~l8_1;
				if (HD_INF_POLICY == 2) {
					// a 16-lane-group pass only for a far match among the 9..16-byte ones, or two and more of them
					const uint64_t bb = b0 | b1;
					if (!((b0 & hbm0) | (b1 & hbm1)) && !(bb & (bb - 1)) && !(b0 && b1))
						b0 = b1 = 0;
				} else if (HD_INF_POLICY) {
					const uint32_t n8 = (uint32_t)__popcll(a0) + (uint32_t)__popcll(a1);
					const uint32_t n16 = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
					const bool run16 = n16 && (((b0 & hbm0) | (b1 & hbm1)) != 0 || n16 >= 3 || n8 + n16 <= 4);
					if (run16 && n8 + n16 <= 4) {
						b0 |= a0;
						b1 |= a1;
						a0 = a1 = 0;
					}
					if (!run16)
						b0 = b1 = 0;
				}
				if (a0 | a1)
					group_pass(std::integral_constant<uint32_t, 8>{}, a0, a1);
				if (b0 | b1)
					group_pass(std::integral_constant<uint32_t, 16>{}, b0, b1);
				done0 = a0 | b0;
				done1 = a1 | b1;
				INF_STAT2(1, __popcll(INF_U64(a0)) + __popcll(INF_U64(a1)));
				INF_STAT2(2, __popcll(INF_U64(b0)) + __popcll(INF_U64(b1)));
			}
#endif
			INF_STAT2(0, __popcll(INF_U64(match0)) + __popcll(INF_U64(match1)));
			INF_STAT2(3, __popcll(INF_U64(simple0 & ~done0)) + __popcll(INF_U64(simple1 & ~done1)));
			INF_STAT2(4, __popcll(INF_U64(hbm0 & ~vec0)) + __popcll(INF_U64(hbm1 & ~vec1)));
			INF_STAT2(5, __popcll(INF_U64(match0 & ~simple0 & ~hbm0)) + __popcll(INF_U64(match1 & ~simple1 & ~hbm1)));
			// the other simple ones, one at a time (two at a time, both reads ahead of both writes, measured 2 % slower)
			for (uint64_t sm = simple0 & ~done0; sm;) {
				const uint32_t m = (uint32_t)__ffsll((unsigned long long)sm) - 1;
				asm("s_bitset0_b64 %0, %1" : "+s"(sm) : "s"(m));       // (sm &= sm - 1 is three scalar instructions)
				const uint32_t mlen = readlane(s0.outlen, m), P = readlane(opos0, m), srcp = readlane(srcl0, m);
				const uint8_t v = L.ring[(srcp + lane) & (RING - 1)];
				L.ring[lane < mlen ? ((P + lane) & (RING - 1)) : RING + lane] = v;
			}
			for (uint64_t sm = simple1 & ~done1; sm;) {
				const uint32_t m = (uint32_t)__ffsll((unsigned long long)sm) - 1;
				asm("s_bitset0_b64 %0, %1" : "+s"(sm) : "s"(m));       // (sm &= sm - 1 is three scalar instructions)
				const uint32_t mlen = readlane(s1.outlen, m), P = readlane(opos1, m), srcp = readlane(srcl1, m);
				const uint8_t v = L.ring[(srcp + lane) & (RING - 1)];
				L.ring[lane < mlen ? ((P + lane) & (RING - 1)) : RING + lane] = v;
			}
			if (pend)
				L.ring[pend_idx] = (uint8_t)pend_v;
			auto copy_general = [&](uint32_t mlen, uint32_t P, uint32_t srcp) {
				const uint32_t moff = P - srcp;
				if (wend - srcp <= RING - 64) {
					// source still in the ring (the literals and the simple matches of the whole
					// window are already in)
					if (moff >= mlen) {
						for (uint32_t i = lane; i < mlen; i += 64)
							L.ring[(P + i) & (RING - 1)] = L.ring[(srcp + i) & (RING - 1)];
					} else {
						const float rcp = 1.0f / (float)moff;
						for (uint32_t i = lane; i < mlen; i += 64) {
							uint32_t q = (uint32_t)((float)i * rcp);
							uint32_t r = i - q * moff;
							r = (int32_t)r < 0 ? r + moff : r;
							r = r >= moff ? r - moff : r;
							L.ring[(P + i) & (RING - 1)] = L.ring[(srcp + r) & (RING - 1)];
						}
					}
				} else {
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
					for (uint32_t i = lane; i < mlen; i += 64)
						L.ring[(P + i) & (RING - 1)] = dst[srcp + i];
				}
			};
			if (scalar_far) {
				far_store();
				while (fa | fb) {
					far_issue();
					far_store();
				}
			}
			for (uint64_t mm = match0 & ~simple0 & ~hbm0; mm;) {
				const uint32_t m = (uint32_t)__ffsll((unsigned long long)mm) - 1;
				asm("s_bitset0_b64 %0, %1" : "+s"(mm) : "s"(m));
				copy_general(readlane(s0.outlen, m), readlane(opos0, m), readlane(srcl0, m));
			}
			for (uint64_t mm = match1 & ~simple1 & ~hbm1; mm;) {
				const uint32_t m = (uint32_t)__ffsll((unsigned long long)mm) - 1;
				asm("s_bitset0_b64 %0, %1" : "+s"(mm) : "s"(m));
				copy_general(readlane(s1.outlen, m), readlane(opos1, m), readlane(srcl1, m));
			}
			pos = wend;
			B += b;
			// the walk stopped in front of a token no window takes (long codeword, end of block): a new
			// window there would come back empty (6 % of all windows did) -- the scalar loop is next
			if ((wm & 64) && !over)
				break;
		}
		// hand the position back to the scalar reader
		dw = B >> 5;
		if ((dw >> 6) != cur_piece) {
			const uint32_t piece = dw >> 6;
			if (piece == cur_piece + 1)
				cw = cw_next;
			else
				cw = load_piece(piece);
			cur_piece = piece;
			cw_next = load_piece(piece + 1);
		}
		bb = 0;
		bc = 0;
		refill();
		bb >>= (B & 31);
		bc -= (B & 31);
		return result;
	};

	int32_t st = HD_OK;
	bool static_loaded = false;
	// The direct tables are built in LDS, then kept in VGPRs for the symbol loop:
	// entry i lives in lane i & 63 of register i >> 6, a lookup is one relative
	// v_mov (s_set_gpr_idx) + v_readlane -- no LDS round trip per symbol.
	typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
	u32x8 LT;
	auto load_tables = [&]() {
#pragma unroll
		for (int r = 0; r < (1 << INF_LT_BITS) / 64; r++)
			LT[r] = L.lit[r * 64 + lane];
	};

	INF_CYC_DECL;
	INF_T0(t_kernel);
	for (;;) {
		INF_T0(t_hdr);
		refill();
		lds_p0 = 0xfffffff0u;                     // header parsing reuses the LDS behind L.comp
		const uint32_t bfinal = (uint32_t)bb & 1;
		const uint32_t btype = ((uint32_t)bb >> 1) & 3;
		bb >>= 3;
		bc -= 3;

		if (btype == 0) {
			// ---- stored: decompress_template.h:234-279 -------------------
			int64_t cbits = (consumed_bits() + 7) & ~(int64_t)7;
			if (cbits > 8 * (int64_t)n) { st = HD_BAD_DATA; break; }
			uint32_t ip = (uint32_t)(cbits >> 3);
			if (n - ip < 4) { st = HD_BAD_DATA; break; }
			seek_byte(ip);
			refill();
			const uint32_t len = (uint32_t)bb & 0xffff, nlen = ((uint32_t)bb >> 16) & 0xffff;
			ip += 4;
			if (len != (~nlen & 0xffff)) { st = HD_BAD_DATA; break; }
			if (len > cap - pos) { st = HD_INSUFFICIENT_SPACE; break; }
			if (len > n - ip) { st = HD_BAD_DATA; break; }
			for (uint32_t done = 0; done < len;) {
				const uint32_t step = len - done < 64 ? len - done : 64;
				if (lane < step)
					L.ring[(pos + lane) & (RING - 1)] = src[ip + done + lane];
				pos += step;
				done += step;
				if (pos - flushed >= HD_PIECE)
					flush_pieces();
			}
			seek_byte(ip + len);
		} else if (btype == 3) {
			st = HD_BAD_DATA;
			break;
		} else {
			uint32_t nlit = 288, noff = 32;
			if (btype == 2) {
				// ---- dynamic header: decompress_template.h:101-232 -------
				refill();
				nlit = 257 + ((uint32_t)bb & 31);
				noff = 1 + (((uint32_t)bb >> 5) & 31);
				const uint32_t npre = 4 + (((uint32_t)bb >> 10) & 15);
				bb >>= 14;
				bc -= 14;
				if (lane < 19)
					L.pre_lens[lane] = 0;
				for (uint32_t i = 0; i < npre; i++) {
					refill();
					if (lane == 0)
						L.pre_lens[k_precode_perm[i]] = (uint8_t)((uint32_t)bb & 7);
					bb >>= 3;
					bc -= 3;
				}
				if (!uniform(build_table<2>(L.pre_lens, 19, L.lit, 7, L.lit_sorted, nullptr, L.cl + INF_T_SCRATCH, lane))) { st = HD_BAD_DATA; break; }
				uint8_t *cl = L.cl;
				uint32_t i = 0, prev = 0;
				bool bad = false;
				while (i < nlit + noff) {
					refill();
					if (consumed_bits() > 8 * (int64_t)n + 64) { bad = true; break; }
					const uint32_t e = uniform(L.lit[(uint32_t)bb & 127]);
					const uint32_t cl_len = e & 15, s = e >> 16;
					bb >>= cl_len;
					bc -= cl_len;
					if (s < 16) {
						if (lane == 0)
							cl[i] = (uint8_t)s;
						prev = s;
						i++;
						continue;
					}
					uint32_t rep, val = 0;
					if (s == 16) {
						if (i == 0) { bad = true; break; }
						rep = 3 + ((uint32_t)bb & 3);
						bb >>= 2; bc -= 2;
						val = prev;
					} else if (s == 17) {
						rep = 3 + ((uint32_t)bb & 7);
						bb >>= 3; bc -= 3;
						prev = 0;
					} else {
						rep = 11 + ((uint32_t)bb & 127);
						bb >>= 7; bc -= 7;
						prev = 0;
					}
					for (uint32_t k = lane; k < rep; k += 64)
						cl[i + k] = (uint8_t)val;
					i += rep;
				}
				if (bad || i != nlit + noff) { st = HD_BAD_DATA; break; }
				static_loaded = false;
				if (!uniform(build_table<1>(cl + nlit, noff, L.off, INF_DT_BITS, L.off_sorted, L.off_count, L.cl + INF_T_SCRATCH, lane)) ||
				    !uniform(build_table<0>(cl, nlit, L.lit, INF_LT_BITS, L.lit_sorted, L.lit_count, L.cl + INF_T_SCRATCH, lane))) {
					st = HD_BAD_DATA;
					break;
				}
			} else if (!static_loaded) {
				// ---- static code: decompress_template.h:297-330 ----------
				uint8_t *cl = L.cl;
				for (uint32_t s = lane; s < 320; s += 64)
					cl[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
				build_table<1>(cl + 288, 32, L.off, INF_DT_BITS, L.off_sorted, L.off_count, L.cl + INF_T_SCRATCH, lane);
				build_table<0>(cl, 288, L.lit, INF_LT_BITS, L.lit_sorted, L.lit_count, L.cl + INF_T_SCRATCH, lane);
				static_loaded = true;
			}

			load_tables();
			INF_T1(0, t_hdr);
			// ---- symbol loop ----------------------------------------------
			for (;;) {
				{
					INF_T0(t_win);
					const uint32_t wr = uniform(run_windows(st));
					INF_T1(1, t_win);
					if (wr)                      // 1: end of block consumed, 2: error
						break;
				}
				INF_T0(t_tok);
				// one token through the fully checked scalar path (stream edges, long codes)
				// one flush site for the whole symbol loop (at most 1023 + 258 bytes pending)
				if (pos - flushed >= HD_PIECE)
					flush_pieces();
				refill();
				if (overrun()) { st = HD_BAD_DATA; break; }
				INF_STAT(2, 1);
				const uint32_t li = (uint32_t)bb & ((1u << INF_LT_BITS) - 1);
				uint32_t e = readlane(LT[li >> 6], li & 63);
				if (((e >> 8) & 3) == K_SLOW) {
					INF_STAT(3, 1);
					const uint32_t sl = uniform(slow_decode(bb, L.lit_count, L.lit_sorted, lane));
					e = litlen_entry(sl & 0xffff, sl >> 16);
				}
				const uint32_t clen = e & 15;
				bb >>= clen;
				bc -= clen;
				const uint32_t kind = (e >> 8) & 3;
				if (kind == K_LIT) {
					if (pos == cap) { st = HD_INSUFFICIENT_SPACE; break; }
					// every lane stores the same byte to the same address: no exec juggling
					L.ring[pos & (RING - 1)] = (uint8_t)(e >> 16);
					pos++;
					INF_T1(2, t_tok);
					continue;
				}
				if (kind == K_EOB) {
					INF_STAT(5, 1);
					INF_T1(2, t_tok);
					break;
				}
				const uint32_t eb = (e >> 4) & 15;
				const uint32_t length = (e >> 16) + ((uint32_t)bb & ((1u << eb) - 1));
				bb >>= eb;
				bc -= eb;
				if (length > cap - pos) { st = HD_INSUFFICIENT_SPACE; break; }
				refill();
				const uint32_t di = (uint32_t)bb & ((1u << INF_DT_BITS) - 1);
				uint32_t d = uniform(L.off[di]);                 // (LDS: one token in 130 comes this way, its copy in four registers cost a wave)
				if (((d >> 8) & 3) == K_SLOW) {
					INF_STAT(4, 1);
					const uint32_t sl = uniform(slow_decode(bb, L.off_count, L.off_sorted, lane));
					d = offset_entry(sl & 0xffff, sl >> 16);
				}
				const uint32_t dlen = d & 15, deb = (d >> 4) & 15;
				bb >>= dlen;
				bc -= dlen;
				const uint32_t offset = (d >> 16) + ((uint32_t)bb & ((1u << deb) - 1));
				bb >>= deb;
				bc -= deb;
				if (offset > pos) { st = HD_BAD_DATA; break; }

				// ---- match copy, 64 bytes per step ------------------------
				if (offset <= INF_NEAR) {
					if (offset >= length) {
						// disjoint: the common case, one pass per 64 bytes
						for (uint32_t i = lane; i < length; i += 64)
							L.ring[(pos + i) & (RING - 1)] = L.ring[(pos - offset + i) & (RING - 1)];
					} else {
						// overlapping (run of period `offset`): source index i mod offset,
						// by a uniform reciprocal; all sources lie before pos
						const float rcp = 1.0f / (float)offset;
						for (uint32_t i = lane; i < length; i += 64) {
							uint32_t q = (uint32_t)((float)i * rcp);
							uint32_t r = i - q * offset;
							r = (int32_t)r < 0 ? r + offset : r;
							r = r >= offset ? r - offset : r;
							L.ring[(pos + i) & (RING - 1)] = L.ring[(pos - offset + r) & (RING - 1)];
						}
					}
				} else {
					// source was flushed long ago: make our own stores visible
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
					for (uint32_t i = lane; i < length; i += 64)
						L.ring[(pos + i) & (RING - 1)] = dst[pos - offset + i];
				}
				pos += length;
				INF_T1(2, t_tok);
			}
			if (st != HD_OK)
				break;
		}
		if (bfinal)
			break;
		if ((a.flags & INF_FLUSHED) && consumed_bits() <= 8 * (int64_t)n && consumed_bits() + 7 >= 8 * (int64_t)n)
			break;
		if (consumed_bits() > 8 * (int64_t)n + 64) { st = HD_BAD_DATA; break; }
	}
	if (st == HD_OK && consumed_bits() > 8 * (int64_t)n)
		st = HD_BAD_DATA;

	uint32_t crcv = 0;
	if (st == HD_OK) {
		// tail: bytes [flushed, pos) leave the ring byte-wise
		for (uint32_t i = flushed + lane; i < pos; i += 64)
			dst[i] = L.ring[i & (RING - 1)];
		if (want_crc) {
			// full 16-byte slots of the tail piece, then the < 16 byte remainder
			const uint32_t piece = flushed / HD_PIECE;
			const uint32_t o = flushed + 16 * lane;
			const bool full = o + 16 <= pos;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (full)
				v = *(const uint4 *)&L.ring[o & (RING - 1)];
			if (pos < 16)
				crc.s = 0;                       // no full slot at all: finish() reseeds
			crc.fold(ct, piece, full, v);
			crcv = crc.finish(ct, lane, pos, &L.ring[(pos & ~15u) & (RING - 1)]);
		}
	}
	if (lane == 0) {
		a.out_len[b] = st == HD_OK ? pos : 0;
		if (a.status) a.status[b] = st;
		if (a.crc) a.crc[b] = crcv;
	}
	INF_T1(3, t_kernel);
	INF_CYC_FLUSH;
}

// throughput form: thousands of streams per launch, 25 workgroups' worth of LDS per CU
__global__ __launch_bounds__(64) void k_inflate(InflateArgs a)
{
	__shared__ InfLds L;
	inflate_stream<INF_RING>(a, L);
}

} // namespace hd
