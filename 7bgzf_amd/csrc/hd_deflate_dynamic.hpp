// hd_deflate_dynamic.hpp -- levels >= 2: the wave parse of hd_deflate_static.hpp
// followed by DYNAMIC Huffman coding (BASELINE config 5, "level-6-like").
//
// Replaces, for BGZF_METHOD=hip2..9, libdeflate_deflate_compress at levels >= 2:
// deflate_compress_greedy / _lazy (lib/libdeflate/deflate_compress.c:2530-2809),
// deflate_make_huffman_code (:1319-1396), the precode / header computation
// (:1483-1631) and the dynamic branch of deflate_flush_block (:1861-2018).
// oracle/hd_deflate_twin.c (deflate_dynamic) is the serial statement of exactly
// this kernel and must produce the same bytes.
//
// One wavefront per block, persistent over a grid-stride loop so that every
// workgroup owns one token slab in HBM:
//   pass 1  parse (same 64-position step, prefix-scan greedy, optional one-lane
//           lazy deferral), tokens -> LDS queue -> slab (64 at a time, coalesced
//           4 B/lane), symbol histograms -> LDS (ds_add_u32)
//   build   litlen/offset/precode code lengths: rank sort by all lanes, then the
//           two-queue merge, depth, overflow and RLE steps on lane 0
//   pass 2  tokens read back 64 at a time, codes looked up in LDS, <= 48 bits
//           per token placed by a DPP prefix sum as two <= 32-bit fields
// A DEFLATE block is closed at the first step boundary with >= 32768 tokens.
// HBM traffic: input once, output once, + 8 B per token for the slab (L2/MALL
// resident in practice).
#pragma once
#include <type_traits>
#include "hd_deflate_static.hpp"

namespace hd {

// (4096 one-MiB members -- sixteen emit wavefronts per CU, which the emit kernel needs: one wavefront writes a member -- or
// 64 Ki BGZF blocks per round)
constexpr uint64_t SPLIT_SCRATCH_BUDGET_WG = (uint64_t)17408 << 20;
constexpr uint32_t DYN_SLAB_TOKENS = HD_DYN_BLOCK_TOKENS + 64;

// the records of the workgroup parse (levels >= HD_WG_LEVEL, hd_deflate_wg.hpp): piece k's tokens from token k * HD_WG_CUT on
// (one per byte at most), then { status, CRC-32 }, then one { tokens, literals, short matches, long matches } per piece
//   [ tokens: cap_tok x u32 | status, crc, pad, pad | piece[max_db] x uint4 ]
__host__ __device__ inline SplitLayout wg_layout(uint32_t max_block)
{
	SplitLayout l;
	l.max_db = (max_block + HD_WG_CUT - 1) / HD_WG_CUT;                    // pieces
	l.cap_tok = l.max_db * HD_WG_CUT;
	l.off_rec = (uint64_t)l.cap_tok * 4;
	l.off_ntok = l.off_rec + 16;                                           // the pieces' records
	l.off_hist = l.off_ntok + (uint64_t)l.max_db * 16;
	l.bytes = l.off_hist;
	return l;
}
inline uint32_t &wg_sub_test()                 // tests (hipdeflate_test_beside): a cap on the blocks of a sub-batch, so that a launch of a few
{                                              // thousand blocks walks the SPAN path (two record buffers, the gates) that 16 GiB walk in the bench
	static uint32_t cap = 0;
	return cap;
}
inline uint32_t wg_sub_batch(uint32_t nblocks, uint32_t split_max)
{
	uint64_t sub = SPLIT_SCRATCH_BUDGET_WG / wg_layout(split_max).bytes;
	if (sub > 65536)
		sub = 65536;
	if (wg_sub_test() && sub > wg_sub_test())
		sub = wg_sub_test();
	if (sub < 1)
		sub = 1;
	return sub < nblocks ? (uint32_t)sub : nblocks;
}
// the emit kernel beside the parse (hd_deflate_wg.hpp launch_wg): behind the first records a SECOND buffer of them when the launch is more
// than one sub-batch, a flag line per block of the launch, a counter per sub-batch, the arrival and hand-out counters
constexpr uint32_t WG_BESIDE_COUNTER_WORDS = 64 + 4096 + 4 * 4096;      // arrival, hand-out | emit wavefronts per CU | per SIMD
constexpr uint32_t WG_BESIDE_FLAG_BLOCKS = 1u << 22;      // launches beyond this many blocks (512 MB of flag lines) run one sub-batch at a time
inline bool wg_beside_span(uint32_t nblocks, uint32_t split_max)
{
	return nblocks > wg_sub_batch(nblocks, split_max) && nblocks <= WG_BESIDE_FLAG_BLOCKS;
}
inline uint64_t wg_beside_bytes(uint32_t nblocks, uint32_t split_max)
{
	const uint32_t sub = wg_sub_batch(nblocks, split_max);
	const bool span = wg_beside_span(nblocks, split_max);
	const uint64_t flagged = span ? nblocks : sub;
	return 256 + (span ? (uint64_t)sub * wg_layout(split_max).bytes + 256 : 0) + (flagged + 1) * 128 + 2 * ((uint64_t)(nblocks / sub + 2) * 4 + 255 & ~(uint64_t)255) +
	       WG_BESIDE_COUNTER_WORDS * 4 + 512;
}
// lat: a latency launch of blocks up to 64 KiB -- room for their staged copies behind the records (hd_deflate_wg.hpp k_stage_in)
inline uint64_t wg_scratch_bytes(uint32_t nblocks, uint32_t split_max, bool lat = false)
{
	const uint32_t sub = wg_sub_batch(nblocks, split_max);
	return (((uint64_t)nblocks * 4 + 15) & ~(uint64_t)15) + (uint64_t)sub * wg_layout(split_max).bytes + 16 +
	       (lat && split_max <= 65536 ? (uint64_t)(sub < 128 ? sub : 128) * 65536 + 256 : wg_beside_bytes(nblocks, split_max));
}

inline uint32_t dynamic_grid(uint32_t nblocks, int level)
{
	// one persistent wave per LDS slot of the level (levels 2-4: 14.5 KiB -> 10 resident per CU
	// -- 11 do not fit, measured --; 5: 16 KiB -> 9; with the two-way tables 6: 21 KiB -> 7; 7: 29 KiB -> 5;
	// 8: 35 KiB -> 4; 9: 60 KiB -> 2); a grid larger than what is resident would run its tail serially
	const uint32_t per_cu = level >= 9 ? 2u : level >= 8 ? 4u : level >= 7 ? 5u : level >= 6 ? 7u : level >= 5 ? 9u : 10u;
	const uint32_t slots = 256u * per_cu;
	return nblocks < slots ? nblocks : slots;
}

// Split path: the tokens of a sub-batch wait in HBM between the parse and the emit launch (2 bytes per
// input byte of the largest block the launch admits, + histograms).  Sub-batches are as large as this
// budget allows (measured on 16 GiB of 0xff00-byte blocks at level 2, 4 B per byte: 4.1 GiB -> 148,
// 8.2 GiB -> 154, 33 GiB -> 158 GB/s), at most 65536 blocks.
constexpr uint64_t SPLIT_SCRATCH_BUDGET = (uint64_t)12672 << 20;
constexpr uint32_t SPLIT_SUB_BATCH_MAX = 65536;

// resident waves of the parse kernel of a level (tests/test_abi.py::test_kernel_resource_budgets)
inline uint32_t parse_slots(int level)
{
	return 256u * (level == 2 ? 18u : level <= 4 ? 12u : level <= 5 ? 10u : level <= 6 ? 8u : level <= 7 ? 5u : level <= 8 ? 4u : 2u);
}

inline uint32_t split_sub_batch(uint32_t nblocks, uint32_t split_max, int level)
{
	const uint64_t per = split_layout(split_max).bytes;
	uint64_t sub = SPLIT_SCRATCH_BUDGET / per;
	if (sub > SPLIT_SUB_BATCH_MAX)
		sub = SPLIT_SUB_BATCH_MAX;
	// Large blocks (1 MiB MiGz members: 40 ms of one wave's time each, ~33 DEFLATE blocks to build one
	// after the other in the emit launch) are better off in the fused kernel, whose persistent grid
	// overlaps all phases of different blocks: measured with 1 MiB text blocks, level 3 72 vs 80 GB/s,
	// level 6 53 vs 66 GB/s.
	if (split_max > (256u << 10) + 65536u)
		return 0;
	if (sub >= nblocks)
		return nblocks;
	// launches are whole rounds of the resident parse waves where the budget allows one; when it allows
	// less than half a round the fused kernel is the faster way again
	const uint32_t slots = parse_slots(level);
	if (sub >= slots)
		sub -= sub % slots;
	else if (sub < slots / 2)
		sub = 0;
	return (uint32_t)sub;                                // 0: fused kernel only
}

// largest block a launch can hold: it must fit its slot at least as stored blocks
inline uint32_t split_max_block(uint64_t out_stride, uint32_t out_cap)
{
	const uint64_t cap = out_stride < out_cap ? out_stride : out_cap;
	return cap > 0x7fffffffu ? 0x7fffffffu : (uint32_t)cap;
}

inline uint64_t fused_scratch_bytes(uint32_t nblocks, int level)
{
	return (uint64_t)dynamic_grid(nblocks, level) * DYN_SLAB_TOKENS * 4;
}

// segments parsed in parts (HD_LAT_PARTS; latency mode): segments per parse + emit launch pair -- all of them unless the
// batch is huge -- and the records they need
inline uint32_t part_sub_batch(uint32_t nsegs, uint32_t parts)
{
	uint64_t sub = SPLIT_SCRATCH_BUDGET / ((uint64_t)parts * part_layout().bytes);
	if (sub > SPLIT_SUB_BATCH_MAX)
		sub = SPLIT_SUB_BATCH_MAX;
	return sub < nsegs ? (uint32_t)sub : nsegs;
}

// parts != 0: the blocks are latency segments parsed in that many parts
inline uint64_t dynamic_scratch_bytes(uint32_t nblocks, uint32_t split_max, int level, uint32_t parts = 0, bool lat = false)
{
	if (level < 2)
		return 0;
	if (level >= HD_WG_LEVEL && !parts)
		return wg_scratch_bytes(nblocks, split_max, lat);
	if (parts)
		return fused_scratch_bytes(nblocks, level) + (uint64_t)nblocks * 4 +
		       (uint64_t)part_sub_batch(nblocks, parts) * parts * part_layout().bytes + 16;
	return fused_scratch_bytes(nblocks, level) + (uint64_t)nblocks * 4 +                 // + the overflow flags
	       (uint64_t)split_sub_batch(nblocks, split_max, level) * split_layout(split_max).bytes + 16;
}

#ifdef HD_EMIT_STATS
// experiment build only: cycles of the emit-only kernel by phase (tools/exp_emit_stats.sh)
__device__ unsigned long long g_emit_stats[32];      // [16, 32): k_emit_wg's block builder, wavefront 0 (hd_emit_wg.hpp)
#define EMIT_T0() const unsigned long long t_ph = EMIT ? clock64() : 0ull
#define EMIT_T(k) do { if (EMIT && lane == 0) atomicAdd(&g_emit_stats[k], clock64() - t_ph); } while (0)
#else
#define EMIT_T0() do { } while (0)
#define EMIT_T(k) do { } while (0)
#endif
#ifdef HD_EMIT_STATS
// ... and of build_code for the litlen alphabet, cumulative from its entry: [8 + k]
#define BUILD_T0() const unsigned long long t_bc = clock64()
#define BUILD_T(k) do { if (nsyms == 288 && lane == 0) atomicAdd(&g_emit_stats[8 + (k)], clock64() - t_bc); } while (0)
#else
#define BUILD_T0() do { } while (0)
#define BUILD_T(k) do { } while (0)
#endif

struct HuffScratch {
	uint32_t freq[288];      // working copy (dummy symbols added)
	uint32_t nf[576];        // node weights: leaves ascending, then internal nodes
	uint16_t parent[576];
	uint16_t order[288];     // symbols by (freq, symbol)
	uint8_t depth[576];
	uint32_t blc[16];
	uint32_t next[16];
};

// Code lengths + canonical (bit-reversed) codewords for one alphabet.
// out[s] = code | len << 16.  Mirrors build_code() of the twin step by step.
// lone: the caller is a wavefront ALONE on its SIMD (k_emit_wg): the register form of the merge below -- fewer instructions, what
// sixteen emit wavefronts per CU want -- is a chain of v_readlane -> scalar compare -> branch that such a wavefront runs at ~670
// cycles per node; the lane-0 form against LDS takes ~210 (measured, tools/exp_emit_wg_stats.py)
// The pointers' types are template parameters: the same source for generic pointers (the emit-only kernels: flat instructions,
// whose extra latency their sixteen wavefronts per CU hide) and for LDS-typed ones (k_emit_wg, hd_emit_wg.hpp: ds_read / ds_write --
// a lone wavefront waits for every one of its LDS round trips, and through flat_load a construction took twice as long).
template <class P>
__device__ __forceinline__ void hs_count(P p)
{
	__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class FP, class OP, class HP>
__device__ __noinline__ void build_code_t(FP freq_in, uint32_t nsyms, uint32_t maxbits, OP out, HP hp, uint32_t lane, bool lone)
{
	auto &h = *hp;
	BUILD_T0();
	// working copy, dummies so that at least two symbols are used
	uint32_t nu = 0;
	for (uint32_t base = 0; base < nsyms; base += 64) {
		const uint32_t s = base + lane;
		const uint32_t f = s < nsyms ? freq_in[s] : 0;
		if (s < nsyms)
			h.freq[s] = f;
		nu += __popcll(__ballot(f != 0));
	}
	if (nu < 2) {
		if (lane == 0) {
			if (nu == 0) {
				h.freq[0] = 1;
				h.freq[1] = 1;
			} else {
				h.freq[h.freq[0] ? 1 : 0] = 1;
			}
		}
		nu = 2;
	}
	// rank sort by (frequency, symbol) over the USED symbols only: they are first compacted, in symbol
	// order, into keys freq << 9 | symbol (all distinct), and a key's rank is the number of smaller keys
	// -- nu * ceil(nu / 64) compares instead of nsyms * ceil(nsyms / 64).  The keys borrow the upper
	// half of the node-weight array, which the merge only reaches when it has consumed them.
	auto *const keys = &h.nf[288];
	{
		uint32_t at = 0;
		for (uint32_t base = 0; base < nsyms; base += 64) {
			const uint32_t s = base + lane;
			const uint32_t fs = s < nsyms ? h.freq[s] : 0;
			const uint64_t m = __ballot(fs != 0);
			if (fs)
				keys[at + __popcll(m & ((1ull << lane) - 1))] = (fs << 9) | s;
			at += (uint32_t)__popcll(m);
		}
	}
	BUILD_T(0);
	for (uint32_t base = 0; base < nu; base += 64) {
		const uint32_t x = base + lane;
		const uint32_t mine = x < nu ? keys[x] : 0xffffffffu;
		uint32_t r = 0;
		for (uint32_t t = 0; t < nu; t++)
			r += keys[t] < mine ? 1u : 0u;
		if (x < nu) {
			h.order[r] = (uint16_t)(mine & 511);
			h.nf[r] = mine >> 9;
		}
	}
	if (lane < 16) {
		h.blc[lane] = 0;
		h.next[lane] = 0;
	}
	BUILD_T(1);
	// The serial part: the two-queue merge and the depths of its nodes.
	nu = uniform(nu);
	if (nu <= 128 && !lone) {
		// Up to 128 used symbols -- every offset and precode alphabet, the litlen alphabet of DNA-like data and most text --:
		// the queues live in REGISTERS, element x in lane x % 64 of register x / 64, and the whole wave runs the loop in
		// step.  A head is two v_readlane and a scalar select, a store is two compare-and-select pairs over all lanes, the
		// loop state is scalar registers and no instruction of the loop waits for LDS: ~50 instructions per node where the
		// LDS form below spends 105 and five round trips (38 of the 65 us an emit wavefront of a latency batch took).
		constexpr uint32_t INF = 0xffffffffu;
		const uint32_t lane1 = lane + 64;
		auto rd2 = [&](uint32_t r0, uint32_t r1, uint32_t x) -> uint32_t {
			const uint32_t v0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, (int)(x & 63));
			const uint32_t v1 = (uint32_t)__builtin_amdgcn_readlane((int)r1, (int)(x & 63));
			return (x & 64) ? v1 : v0;
		};
		auto wr2 = [&](uint32_t &r0, uint32_t &r1, uint32_t x, uint32_t v) {
			r0 = lane == x ? v : r0;
			r1 = lane1 == x ? v : r1;
		};
		uint32_t LW0 = lane < nu ? h.nf[lane] : INF, LW1 = lane1 < nu ? h.nf[lane1] : INF;     // leaf weights, ascending
		uint32_t NW0 = 0, NW1 = 0;           // node weights, in the order the nodes are made (node m: the twin's k = nu + m)
		uint32_t LP0 = 0, LP1 = 0, NP0 = 0, NP1 = 0;                                           // parents, as node numbers
		// an empty queue's head weighs INF, so "take the leaf" is one compare (ties: the leaf first); a freshly made
		// node that finds the node queue empty is its head
		uint32_t i = 0, j = 0, m = 0;
		uint32_t lv = rd2(LW0, LW1, 0), lv2 = nu > 1 ? rd2(LW0, LW1, 1) : INF, iv = INF;
		while (m + 1 < nu) {
			uint32_t sum = 0;
#pragma unroll
			for (int t = 0; t < 2; t++) {
				if (lv <= iv) {
					sum += lv;
					wr2(LP0, LP1, i, m);
					i++;
					lv = lv2;
					lv2 = i + 1 < nu ? rd2(LW0, LW1, i + 1) : INF;
				} else {
					sum += iv;
					wr2(NP0, NP1, j, m);
					j++;
					iv = j < m ? rd2(NW0, NW1, j) : INF;
				}
			}
			wr2(NW0, NW1, m, sum);
			if (iv == INF)
				iv = sum;
			m++;
		}
		// depths of the nodes, root (the last one made: depth 0) first: a parent has the larger number
		uint32_t ND0 = 0, ND1 = 0;
		for (uint32_t x = nu - 2; x-- > 0;)
			wr2(ND0, ND1, x, rd2(ND0, ND1, rd2(NP0, NP1, x)) + 1);
		// the leaves' depths, through LDS: a lane's parent is any node
		if (lane + 1 < nu)
			h.depth[lane] = (uint8_t)ND0;
		if (lane1 + 1 < nu)
			h.depth[lane1] = (uint8_t)ND1;
		if (lane < nu) {
			const uint32_t d = (uint32_t)h.depth[LP0] + 1;
			hs_count(&h.blc[d > maxbits ? maxbits : d]);
		}
		if (lane1 < nu) {
			const uint32_t d = (uint32_t)h.depth[LP1] + 1;
			hs_count(&h.blc[d > maxbits ? maxbits : d]);
		}
	} else {
		// The serial parts run on lane 0 against LDS, ~100 cycles a round trip, so they are written to keep
		// as few loads as possible on the critical path (same algorithm and results as the twin's loops).
		if (lane == 0) {
			// two-queue merge: leaves 0..nu-1 ascending, internal nodes nu..2nu-2.  The heads of both
			// queues are held in registers (the leaf queue one element ahead), a node's weight is the
			// sum of two register values, and a freshly made node that is the internal head is taken
			// from the register it was computed in.
			uint32_t i = 0, j = nu, k = nu;
			uint32_t lv = h.nf[0], lv2 = nu > 1 ? h.nf[1] : 0u;     // leaf head and the one behind it
			uint32_t iv = 0;                                      // internal head, valid while j < k
			while (k < 2 * nu - 1) {
				uint32_t sum = 0;
	#pragma unroll
				for (int t = 0; t < 2; t++) {
					if (i < nu && (j >= k || lv <= iv)) {
						sum += lv;
						h.parent[i] = (uint16_t)k;
						i++;
						lv = lv2;
						lv2 = i + 1 < nu ? h.nf[i + 1] : 0u;
					} else {
						sum += iv;
						h.parent[j] = (uint16_t)k;
						j++;
						iv = j < k ? h.nf[j] : 0u;
					}
				}
				h.nf[k] = sum;
				if (j == k)
					iv = sum;
				k++;
			}
			// depths of the internal nodes, root first (a parent has the larger index); the next
			// node's parent index is loaded while this one's depth is on its way
			h.depth[2 * nu - 2] = 0;
			if (nu > 2) {
				uint32_t pn = h.parent[2 * nu - 3];
				for (int x = (int)(2 * nu - 3); x >= (int)nu; x--) {
					const uint32_t pc = pn;
					if (x > (int)nu)
						pn = h.parent[x - 1];
					h.depth[x] = (uint8_t)(h.depth[pc] + 1);
				}
			}
		}
		// leaves: depth and level counts by all lanes; leaves deeper than maxbits are cut back to maxbits
		for (uint32_t base = 0; base < nu; base += 64) {
			const uint32_t x = base + lane;
			if (x < nu) {
				const uint32_t d = (uint32_t)h.depth[h.parent[x]] + 1;
				hs_count(&h.blc[d > maxbits ? maxbits : d]);
			}
		}
	}
	BUILD_T(2);
	if (lane == 0) {
		// The cut leaves the code over-subscribed by `excess` codewords of length maxbits (Kraft sum in
		// units of 2^-maxbits); every pass gives one back: a leaf moves one level down, a maxbits leaf
		// becomes its sibling.  (Counting the cut leaves is only right for leaves at maxbits + 1.)
		int32_t excess = -(int32_t)(1u << maxbits);
		for (uint32_t bits = 1; bits <= maxbits; bits++)
			excess += (int32_t)(h.blc[bits] << (maxbits - bits));
		while (excess > 0) {
			uint32_t bits = maxbits - 1;
			while (bits >= 1 && h.blc[bits] == 0)
				bits--;
			if (bits == 0)
				break;                          // only with more symbols than 2^maxbits: no such code exists
			h.blc[bits]--;
			h.blc[bits + 1] += 2;
			h.blc[maxbits]--;
			excess--;
		}
		// canonical first codes
		uint32_t code = 0;
		for (uint32_t bits = 1; bits <= maxbits; bits++) {
			code = (code + (bits > 1 ? h.blc[bits - 1] : 0)) << 1;
			h.next[bits] = code;
		}
	}
	BUILD_T(3);
	// lengths: the leaf at sorted position x gets `bits` where the level counts,
	// walked from maxbits down, reach x  (smallest frequency = longest code)
	for (uint32_t base = 0; base < nsyms; base += 64)
		if (base + lane < nsyms)
			out[base + lane] = 0;
	for (uint32_t base = 0; base < nu; base += 64) {
		const uint32_t x = base + lane;
		uint32_t acc = 0, bits = 0;
		for (uint32_t bb = maxbits; bb >= 1; bb--) {
			const uint32_t c = h.blc[bb];
			if (bits == 0 && x < acc + c)
				bits = bb;
			acc += c;
		}
		if (x < nu)
			out[h.order[x]] = bits << 16;
	}
	BUILD_T(4);
	// canonical codewords in symbol order: next[len] + rank among equal lengths
	uint32_t run[16];
#pragma unroll
	for (int l = 0; l < 16; l++)
		run[l] = 0;
	for (uint32_t base = 0; base < nsyms; base += 64) {
		const uint32_t s = base + lane;
		const uint32_t len = s < nsyms ? out[s] >> 16 : 0;
		uint32_t rank = 0;
#pragma unroll
		for (int l = 1; l < 16; l++) {
			const uint64_t m = __ballot(len == (uint32_t)l);
			if (len == (uint32_t)l)
				rank = run[l] + __popcll(m & ((1ull << lane) - 1));
			run[l] += __popcll(m);
		}
		if (len) {
			const uint32_t cw = h.next[len] + rank;
			out[s] = (len << 16) | (__brev(cw) >> (32 - len));
		}
	}
	BUILD_T(5);
}

// (the form the emit-only kernels call)
__device__ __forceinline__ void build_code(const uint32_t *freq_in, uint32_t nsyms, uint32_t maxbits, uint32_t *out, HuffScratch &h,
					    uint32_t lane, bool lone = false)
{
	build_code_t<const uint32_t *, uint32_t *, HuffScratch *>(freq_in, nsyms, maxbits, out, &h, lane, lone);
}

struct DynLds {
	uint32_t lf[288], df[32];          // histograms of the open DEFLATE block
	uint32_t pcode[19], pfreq[19];
	uint32_t misc[8];                  // 0: #items  1: hlit  2: hdist  3: hclen  4: PARTS: the segment's CRC-32 (second wavefront)
};

// Scratch of the code construction.  It is only live while a DEFLATE block is being
// closed, when the parse is paused, so it shares its LDS with the ring window (the
// ring is re-read from HBM after a non-final block; 7 KiB less LDS per wave).
struct DynBuild {
	HuffScratch hs;
	// The RLE of the code lengths (items: symbol | extra << 8) and the lengths themselves are written
	// after the litlen and offset codes are built and read until the header is out; in between only the
	// 19-symbol precode is built, which touches the first 38 nodes of hs.nf -- so they live further up in
	// that array instead of costing 960 bytes of their own (the emit-only kernel: 8 LDS units, 16 waves)
	__device__ __forceinline__ uint16_t *items() { return (uint16_t *)&hs.nf[64]; }       // 320 x u16
	__device__ __forceinline__ uint8_t *lens() { return (uint8_t *)&hs.nf[64 + 160]; }    // 320 x u8
	uint32_t lcode[288], dcode[32];    // code | len << 16: live from the construction to the end of the emit pass
};

__constant__ uint8_t k_perm19[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

// EMIT = 1: emit-only mode, the second half of the level-2 split path: tokens, histograms and
// CRC of blocks [a.first, a.first + a.count) are in the scratch (written by k_deflate_static<.., true>),
// this kernel builds the codes and writes the members exactly as the fused mode would have.
// PARTS (EMIT only): the tokens of a block lie in PARTS records, one per parse part (hipdeflate_params.h HD_LAT_PARTS)
// ... and the workgroup has a SECOND wavefront that builds the offset code while the first builds the litlen code (an emit
// wavefront of a latency batch is alone on its CU: the two constructions one behind the other were 30 of its 55 us)
// BESIDE (EMIT only, the workgroup levels): the instantiation that runs beside the parse (hd_deflate_wg.hpp launch_wg) -- blocks handed out
// by a counter, a wait for the block's flag, an acquire behind it; BESIDE = 0 is the kernel of rounds 4-5 to the instruction
// CRC-32 of a whole block by one wavefront (the rare paths of the workgroup levels' emit kernel: a block its parse workgroup gave up --
// a stall, a poisoned gate -- is written stored, and the parse's own CRC stopped where its filler did)
__device__ __noinline__ uint32_t crc_of_block(const CrcTables *ct, const uint8_t *src, uint32_t n, uint32_t lane)
{
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	CrcLanes crc;
	crc.init(lane, n);
	const uint32_t np = (n + HD_PIECE - 1) / HD_PIECE;
	for (uint32_t k = 0; k < np; k++)
		crc.fold<true>(ct, k, k * HD_PIECE + 16 * lane + 16 <= n, load_slot(src, n, k, lane, aligned));      // (the chained form: few registers -- this is a call from a kernel at its budget)
	return crc.finish(ct, lane, n, src + (n & ~15u));
}
// (a pointer known to be to device memory: the records' loads are global_load, not flat_load -- a flat load counts as an LDS access too,
// and every wait for the LDS in the token loops would wait for the tokens requested ahead)
typedef const __attribute__((address_space(1))) uint32_t *hd_global_u32p;
#ifndef HD_BESIDE_SC1_LOADS
#define HD_BESIDE_SC1_LOADS 1
#endif
#ifndef HD_EXP_NO_COUNT
#define HD_EXP_NO_COUNT 0
#endif
#ifndef HD_BESIDE_KEEP
#define HD_BESIDE_KEEP 3                         // emit wavefronts a CU keeps beside a parse workgroup (k_deflate_dynamic<..., BESIDE>)
#endif
#ifndef HD_BESIDE_EMIT_PRIO
#define HD_BESIDE_EMIT_PRIO 0
#endif
template <int WIN_BITS, int HASH_BITS, int MINLEN, int LAZY, int EMIT, int INTRA = 0, int DEEP = 0, int PARTS = 0, int BESIDE = 0>
__global__ __launch_bounds__(PARTS ? 128 : 64) void k_deflate_dynamic(DeflateArgs a)
{
	static_assert(!BESIDE || (EMIT && !PARTS), "beside the parse runs the emit-only kernel of the workgroup levels");
	static_assert(!PARTS || EMIT, "parts are a matter of the emit-only kernel");
	constexpr uint32_t W = 1u << WIN_BITS;
	constexpr uint32_t W4M = W / 4 - 1;
	// DEEP (the lazy levels, hipdeflate_params.h "LAZY LEVELS"): dword buckets of two positions; HS counts 16-bit units
	constexpr uint32_t NB = HD_BUCKETS(WIN_BITS, HASH_BITS);
	constexpr uint32_t HS = DEEP ? 2 * NB : HD_TABLE_ENTRIES(WIN_BITS, HASH_BITS);
	// staging ring (dwords) and flush granule: the token loop adds up to 64 x 48 bits = 96 dwords to
	// fewer than FLUSH_DW pending ones before it flushes one granule, so 128 + 96 <= 256 is what it takes
	// (a 128-dword ring would do for typical data and overflow on 48-bit tokens)
	constexpr uint32_t STG = 256;
	constexpr uint32_t FLUSH_DW = STG / 2;

	__shared__ __attribute__((aligned(16))) union {
		uint32_t ring[EMIT ? 4 : W / 4 + 4];
		DynBuild build;
	} U;
	// (the union is as large as the larger of the two: at level 2 the 4 KiB ring sits inside the scratch)
	uint32_t *const ring32 = U.ring;
	DynBuild &Bd = U.build;
	__shared__ __attribute__((aligned(16))) uint16_t table[EMIT ? 8 : HS + 8];
	__shared__ __attribute__((aligned(16))) uint32_t stage[STG];
	__shared__ DynLds L;
	// token queue: < 64 waiting + <= 64 of one step.  (No dump slots for the lanes without a token as in
	// the level-1 kernel: LDS is granted in 1280-byte units and levels 2-4 must stay within 12 of them
	// for 10 waves per CU.)
	constexpr uint32_t TOKQ = 128;
	__shared__ uint32_t tokq[EMIT ? 1 : TOKQ];
	const uint8_t *ring8 = (const uint8_t *)ring32;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t b_end = (BESIDE && a.span_sub) ? a.nblocks : EMIT ? (a.first + a.count < a.nblocks ? a.first + a.count : a.nblocks) : a.nblocks;
	__shared__ typename std::conditional<(PARTS > 0), HuffScratch, uint32_t>::type hs2;   // the second wavefront's construction scratch
	if (PARTS && threadIdx.x >= 64) {
		// PARTS: the second wavefront.  It walks the same blocks as the first and meets it at two barriers per block -- the
		// histograms are in L.df / the offset code is in Bd.dcode (flush_block) -- and does nothing else
		for (uint32_t b = a.first + blockIdx.x; b < b_end; b += gridDim.x) {
			__syncthreads();
			build_code(L.df, 32, HD_OFFSET_MAXBITS, Bd.dcode, *(HuffScratch *)&hs2, lane);
			{
				// ... and the segment's CRC-32 from its parts': lane q moves part q's CRC to the end of the segment (the
				// bytes behind the part appended: a chain of dependent table loads), the lanes are XOR-ed
				const uint32_t n = a.in_len[b];
				const SplitLayout lay = part_layout();
				const uint8_t *rec0 = part_block(a.scratch, (b - a.first) * PARTS);
				const uint32_t np = (n + HD_LAT_PART_BYTES - 1) / HD_LAT_PART_BYTES;
				uint32_t c = 0;
				if (lane < np) {
					const uint32_t end = (lane + 1) * HD_LAT_PART_BYTES < n ? (lane + 1) * HD_LAT_PART_BYTES : n;
					c = crc_append_bytes(a.ct, ((const uint32_t *)(rec0 + (uint64_t)lane * lay.bytes + lay.off_rec))[1], n - end);
				}
				c = wave_xor_reduce(c);
				if (lane == 0)
					L.misc[4] = c;
			}
			__syncthreads();
		}
		return;
	}
	const ClockStamp clk(HD_CLK_DYNAMIC);
	if constexpr (BESIDE != 0) {
		if (a.arrived) {
			// A candidate for a place beside the parse.  A parse workgroup fits a CU whose LDS is free from 30 KB up and whose SIMDs hold at
			// most ONE of us each (4 x 96 registers + ours of a SIMD's 512).  More candidates than places are launched -- other kernels
			// are about when this one starts (the caller's gather of the pass before), and 768 wavefronts simply trusted to land three
			// to a CU did not always: a parse a quarter slower, on and off -- and a candidate STAYS if its LDS block is one of the three
			// lowest of its CU (HW_REG_LDS_ALLOC: base in 256-byte units in the low bits, size in [20:12]; tools/beside_filter_probe.hip) and
			// no other has its SIMD; the others leave at once, and what they free lies ABOVE the ones that stay (kept by order of
			// arrival instead, the leavers left holes and the parse's 131 KB did not fit: measured, three times slower).  The blocks
			// are handed out by a counter, so nobody is missed.  (a.arrived counts everybody who has decided: the gate waits for it.)
			bool stay = true;
			if (lane == 0) {
				unsigned hw, xcc, la;
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
				// cu: xcc_id [3:0] | se_id [15:13] | sh_id [12] | cu_id [11:8] of HW_ID -- 12 bits; simd_id [5:4]
				const uint32_t cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
				uint32_t *const places = a.arrived + 64;
				stay = (la & 0xfffu) < (a.beside_keep < (uint32_t)HD_BESIDE_KEEP ? a.beside_keep : (uint32_t)HD_BESIDE_KEEP) * ((la >> 12) & 0x1ffu) &&
				       __hip_atomic_fetch_add(&places[4096 + 4 * cu + ((hw >> 4) & 3u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
				__hip_atomic_fetch_add(a.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (!uniform((uint32_t)stay))
				return;
#if HD_BESIDE_EMIT_PRIO
			__builtin_amdgcn_s_setprio(HD_BESIDE_EMIT_PRIO);
#endif
		}
	}
	uint32_t *tok = (uint32_t *)a.scratch + (uint64_t)blockIdx.x * DYN_SLAB_TOKENS;
	const CrcTables *ct = a.ct;

	// (beside the parse, hd_deflate_wg.hpp launch_wg: the blocks are handed out by a counter -- the resident wavefronts and the ones that
	// follow the parse take from the same one --; elsewhere wavefront j has blocks j, j + grid, ...)
	uint32_t cur_sub = (BESIDE && a.take_sub) ? a.take_sub - 1 : 0u;      // (SPAN: the sub-batch this wavefront is taking blocks of)
	auto take = [&](uint32_t b_now, bool first_one) -> uint32_t {
		if constexpr (BESIDE != 0) {
			uint32_t t = 0;
			if (lane == 0) {
				// (SPAN: the member of b_now is done -- every read of its records has returned: the sub-batch's count)
				if (!first_one && a.span_sub)
					__hip_atomic_fetch_add(&a.emitted[b_now / a.span_sub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (a.span_sub) {
					// SPAN: a hand-out counter per sub-batch, taken in order.  A counter that has run past its sub-batch's blocks sends the
					// wavefront on to the next one (what it added there is lost on nobody: there is no such block) -- so the launch in front
					// of a gate (take_sub: the ONE sub-batch it serves) takes what is left of its sub-batch with one fetch_add per wavefront
					// and cannot take a block of another (a shared counter bounded by compare-and-swap was tried: 4096 wavefronts that find
					// something left retry against each other, 25 ms)
					uint32_t k = cur_sub;
					const uint32_t k_end = a.take_sub ? a.take_sub : (a.nblocks + a.span_sub - 1) / a.span_sub;
					t = 0xffffffffu;
					while (k < k_end) {
						const uint32_t left = a.nblocks - k * a.span_sub, cntk = left < a.span_sub ? left : a.span_sub;
						const uint32_t v = __hip_atomic_fetch_add(&a.next[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						if (v < cntk) {
							t = k * a.span_sub + v;
							break;
						}
						k++;
					}
					cur_sub = k;
				} else {
					t = __hip_atomic_fetch_add(a.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			cur_sub = uniform(cur_sub);
			return (a.span_sub ? 0u : a.first) + uniform(t);
		} else {
			return first_one ? (EMIT ? a.first : 0u) + blockIdx.x : b_now + gridDim.x;
		}
	};
	for (uint32_t b = take(0, true); b < b_end; b = take(b, false)) {
		const uint8_t *src = a.in + a.in_off[b];
		const uint32_t n = a.in_len[b];
		if (PARTS ? false : EMIT ? (!a.wg && a.split_ovf[b] != 0) : (a.skip_small && a.split_ovf[b] == 0))
			continue;                            // the other path's block
		if (!EMIT && a.seg_limit && n > a.seg_limit)
			continue;                            // coded in segments (hd_segment.hpp)
		const bool aligned = (((uintptr_t)src) & 15) == 0;
		uint32_t *dst32 = (uint32_t *)(a.out + (uint64_t)b * a.out_stride);

		const uint32_t hdr = frame_hdr_bytes(a.frame), trl = frame_trl_bytes(a.frame);
		uint64_t cap64 = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
		if (a.frame == HD_FRAME_BGZF && cap64 > 65536)
			cap64 = 65536;
		const uint32_t cap = (uint32_t)cap64;
		const uint32_t stored = HD_STORED_SIZE(n);
		uint32_t limit = stored - 1;
		const bool flush = a.frame == HD_FRAME_RAW_FLUSH;
		const uint32_t sfx = frame_sfx_bytes(a.frame);
		bool alive = cap >= hdr + trl + sfx + 2;
		if (alive && cap - hdr - trl - sfx < limit)
			limit = cap - hdr - trl - sfx;
		const uint64_t limit_bits = 8ull * limit;

		// ---- init LDS ---------------------------------------------------
		if (!EMIT) {
			for (uint32_t i = lane; i < HS / 8 + 1; i += 64)
				((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);
			for (uint32_t i = lane; i < 288; i += 64)
				L.lf[i] = 0;
			if (lane < 32)
				L.df[lane] = 0;
		}
		for (uint32_t i = lane; i < STG; i += 64)
			stage[i] = 0;
		if (lane < 4 && hdr)
			stage[lane] = frame_hdr_word(a.frame, lane);

		CrcLanes crc;
		crc.init(lane, n);
		uint32_t filled = 0;
		uint4 pre = EMIT ? make_uint4(0, 0, 0, 0) : load_slot(src, n, 0, lane, aligned);
		uint32_t bitpos = 8 * hdr, flushed = 0;
		const uint32_t paybase = 8 * hdr;
		uint32_t ntok = 0;                   // tokens of the open DEFLATE block: in the slab + still queued
		uint32_t ntok_slab = 0, qhead = 0, qtail = 0;
		// PARTS: token k of the block is tok[k + sum of pgap[q] over the parts q >= 1 that start at or before k]
		// (pcum[q] = tokens ahead of part q; pgap[q] = what is left of part q - 1's slab behind its tokens)
		uint32_t pcum[PARTS ? PARTS : 1], pgap[PARTS ? PARTS : 1];
		auto tix = [&](uint32_t k) -> uint32_t {
			uint32_t o = k;
#pragma unroll
			for (int q = 1; q < PARTS; q++)
				o += k >= pcum[q] ? pgap[q] : 0u;
			return o;
		};

		auto put = [&](uint32_t code, uint32_t nbits, uint32_t bp) {
			if (nbits) {
				const uint32_t sh = bp & 31, i = (bp >> 5) & (STG - 1);
				atomicOr(&stage[i], code << sh);
				if (sh + nbits > 32)
					atomicOr(&stage[(i + 1) & (STG - 1)], code >> (32 - sh));
			}
		};
		auto flush_ready = [&]() {
			if ((bitpos >> 5) - flushed >= FLUSH_DW) {
				if (FLUSH_DW == 128) {
					const uint32_t i = (flushed & (STG - 1)) + 2 * lane;   // flushed is a multiple of FLUSH_DW
					const uint2 v = *(const uint2 *)&stage[i];
					*(uint2 *)&stage[i] = make_uint2(0, 0);
					*(uint2 *)&dst32[flushed + 2 * lane] = v;
				} else {
					const uint32_t i = (flushed & (STG - 1)) + lane;
					const uint32_t v = stage[i];
					stage[i] = 0;
					dst32[flushed + lane] = v;
				}
				flushed += FLUSH_DW;
			}
		};
		// one field per lane (nbits <= 32, 0 = none), in lane order
		auto emit1 = [&](uint32_t code, uint32_t nbits) {
			const uint32_t incl = wave_incl_scan(nbits);
			put(code, nbits, bitpos + incl - nbits);
			bitpos += readlane(incl, 63);
			flush_ready();
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
				((uint4 *)ring32)[W / 16] = v;
			crc.template fold<true>(ct, piece, piece * HD_PIECE + lane * 16 + 16 <= n, v);
		};
		struct Fetched {
			uint32_t v, vh, c, c2;
		};
		auto fetch = [&](uint32_t S_) -> Fetched {
			Fetched f;
			const uint32_t p = S_ + lane;
			const uint32_t *w = &ring32[(p >> 2) & W4M];
			const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
			f.v = __builtin_amdgcn_alignbyte(w1, w0, p & 3);
			f.vh = __builtin_amdgcn_alignbyte(w2, w1, p & 3);
			const bool can = p + (DEEP ? HD_LAZY_KEY_BYTES : HD_MIN_MATCH) <= n;
			uint32_t e, e2 = 0;
			if (DEEP) {
				// (a lane past the end reads and writes the spare bucket behind the table)
				uint32_t *const bk = (uint32_t *)table;
				const uint32_t h = can ? HD_HASH_SLOT6(f.v, f.vh, NB) : NB;
				const uint32_t eb = bk[h];
				bk[h] = (eb << 16) | ((p + 1) & 0xffffu);
				e = eb & 0xffffu;
				e2 = eb >> 16;
			} else {
				const uint32_t h = can ? HD_HASH_SLOT(f.v, HS) : HS;
				const uint16_t mine = (uint16_t)(p + 1);
				e = table[h];
				table[h] = mine;
			}
			const uint32_t back = (p + 1 - e) & 0xffffu;
			f.c = (can && e && back) ? p + 1 - back : 0u;
			const uint32_t back2 = (p + 1 - e2) & 0xffffu;
			f.c2 = (DEEP && can && e2 && back2) ? p + 1 - back2 : 0u;
			if (INTRA) {
				// a nearer occurrence inside the step replaces the table's candidate
				const uint32_t d = intra_step_distance<INTRA>(f.v, lane);
				f.c = (can && d) ? p + 1 - d : f.c;
			}
			// (lanes that hash alike stored to one entry in that one instruction: the highest lane stays,
			// hd_deflate_static.hpp fetch())
			return f;
		};
		struct Probed {
			uint32_t c0, c1, c2;
		};
		auto probe = [&](uint32_t c) -> Probed {
			Probed q;
			const uint32_t *w = &ring32[((c - 1) >> 2) & W4M];
			q.c0 = w[0];
			q.c1 = w[1];
			q.c2 = w[2];
			return q;
		};

		// ---- the workgroup parse's records (a.wg; hd_deflate_wg.hpp): the DEFLATE block being closed is the pieces
		// [wg_k0, wg_k1) of the block; piece k's tokens start at token k * HD_WG_CUT of the record ------------------------
		uint32_t wg_k0 = 0, wg_k1 = 0, wg_base = 0xffffff00u, wg_np = 0;
		const uint4 *wg_pieces = nullptr;
		uint4 wg_pv = make_uint4(0, 0, 0, 0);            // lane i: the record of piece wg_base + i
		auto wg_piece_load = [&](uint32_t k) {
			if (k - wg_base >= 64) {
				wg_base = k & ~63u;
				if (BESIDE && HD_BESIDE_SC1_LOADS) {
					// (beside the parse: what its workgroups wrote is read by loads that are coherent at the device's level themselves -- sc1 --
					// instead of behind an invalidate of this XCD's L2 per block; see the wait below)
					wg_pv = make_uint4(0, 0, 0, 0);
					if (wg_base + lane < wg_np) {
						const hd_global_u32p pp = (hd_global_u32p)(const uint32_t *)&wg_pieces[wg_base + lane];
						wg_pv.x = __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						wg_pv.y = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						wg_pv.z = __hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						wg_pv.w = __hip_atomic_load(pp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				} else
				wg_pv = wg_base + lane < wg_np ? wg_pieces[wg_base + lane] : make_uint4(0, 0, 0, 0);
			}
		};
		auto wg_piece_tokens = [&](uint32_t k) -> uint32_t {
			wg_piece_load(k);
			return readlane(wg_pv.x, k - wg_base);
		};
		// f(token of lane i, tokens in the group) for every group of up to 64 tokens of the DEFLATE block, in order; the
		// loads run a batch of eight groups ahead of their use (see the note at the emit-only kernel's token loop)
		auto wg_for_tokens = [&](auto &&f) {
			constexpr uint32_t PF = 8;
			uint32_t k = wg_k0, base = 0, c = wg_k0 < wg_k1 ? wg_piece_tokens(wg_k0) : 0u;
			uint32_t cur[PF], nxt[PF], ncur[PF], nnxt[PF];
			auto fetch = [&](uint32_t &v, uint32_t &nv) {
				nv = 0;
				v = 0;
				if (k < wg_k1) {
					nv = c - base < 64 ? c - base : 64;
					if (BESIDE && HD_BESIDE_SC1_LOADS)
						v = lane < nv ? __hip_atomic_load((hd_global_u32p)tok + (k * HD_WG_CUT + base + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
					else
					v = lane < nv ? tok[k * HD_WG_CUT + base + lane] : 0u;
					base += 64;
					if (base >= c) {
						k++;
						base = 0;
						c = k < wg_k1 ? wg_piece_tokens(k) : 0u;
					}
				}
			};
#pragma unroll
			for (uint32_t j = 0; j < PF; j++)
				fetch(nxt[j], nnxt[j]);
			while (nnxt[0]) {
#pragma unroll
				for (uint32_t j = 0; j < PF; j++) {
					cur[j] = nxt[j];
					ncur[j] = nnxt[j];
				}
#pragma unroll
				for (uint32_t j = 0; j < PF; j++)
					fetch(nxt[j], nnxt[j]);
#pragma unroll
				for (uint32_t j = 0; j < PF; j++)
					if (ncur[j])
						f(cur[j], ncur[j]);
			}
		};

		// Close the open DEFLATE block: build codes, pick dynamic vs static, emit.
		// Returns false if the member would exceed `limit` (-> stored fallback).
		auto flush_block = [&](bool final) -> bool {
			if (lane == 0)
				L.lf[256] += 1;                     // end of block
			{
				EMIT_T0();
				if (PARTS)
					__syncthreads();                    // the second wavefront starts on the offset code
				build_code(L.lf, 288, HD_LITLEN_MAXBITS, Bd.lcode, Bd.hs, lane);
				if (PARTS)
					__syncthreads();                    // ... and has it
				else
					build_code(L.df, 32, HD_OFFSET_MAXBITS, Bd.dcode, Bd.hs, lane);
				EMIT_T(0);
			}
			EMIT_T0();
			if (lane < 19)
				L.pfreq[lane] = 0;
			// ---- code lengths and their RLE (deflate_compute_precode_items, deflate_compress.c:1483-1557),
			// by all lanes: on lane 0 this loop was a fifth of the emit kernel's time (~300 dependent LDS
			// round trips).  hlit / hdist from two ballots; run starts by ballot + popcount per 64 lengths;
			// one lane per run computes how many 18 / 17 / 16 / literal items the greedy rule makes of it, a
			// prefix sum places them, the lane writes them.
			uint32_t hlit, hdist;
			{
				const uint64_t ml = __ballot(lane < 29 && (Bd.lcode[257 + lane] >> 16) != 0);
				const uint64_t md = __ballot(lane < 29 && (Bd.dcode[1 + lane] >> 16) != 0);
				hlit = ml ? 257 + 64 - (uint32_t)__clzll((long long)ml) : 257;
				hdist = md ? 1 + 64 - (uint32_t)__clzll((long long)md) : 1;
			}
			const uint32_t total = hlit + hdist;
			uint8_t *lens = Bd.lens();
			uint16_t *items = Bd.items();
			uint16_t *run_start = Bd.hs.parent;            // free until the precode is built
			for (uint32_t i = lane; i < total; i += 64)
				lens[i] = (uint8_t)((i < hlit ? Bd.lcode[i] : Bd.dcode[i - hlit]) >> 16);
			uint32_t nruns = 0;
			for (uint32_t base = 0; base < total; base += 64) {
				const uint32_t i = base + lane;
				const bool st = i < total && (i == 0 || lens[i] != lens[i - 1]);
				const uint64_t m = __ballot(st);
				if (st)
					run_start[nruns + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)i;
				nruns += (uint32_t)__popcll(m);
			}
			if (lane == 0)
				run_start[nruns] = (uint16_t)total;
			uint32_t ni = 0;
			for (uint32_t rb = 0; rb < nruns; rb += 64) {
				const uint32_t r = rb + lane;
				const bool valid = r < nruns;
				const uint32_t s0 = valid ? run_start[r] : 0, len = valid ? run_start[r + 1] - s0 : 0;
				const uint32_t v = valid ? lens[s0] : 0;
				// the greedy rule in closed form: `big` full-size repeat items, one more for a rest that is
				// long enough, single symbols for what is left (and in front, for a non-zero length)
				uint32_t rep, big, rest, lead;
				if (v == 0) {
					rep = 18; lead = 0;
					big = len / 138; rest = len - 138 * big;
				} else {
					rep = 16; lead = valid ? 1u : 0u;
					big = (len - lead) / 6; rest = (len - lead) - 6 * big;
				}
				const uint32_t full = v == 0 ? 138u : 6u, base_len = v == 0 ? 11u : 3u;
				uint32_t extra_rep = 0, extra_sym = rep, extra_base = base_len;   // the one item for the rest
				if (rest >= base_len) {
					extra_rep = 1;
				} else if (v == 0 && rest >= 3) {
					extra_rep = 1; extra_sym = 17; extra_base = 3;
				}
				const uint32_t tail = extra_rep ? 0u : rest;
				const uint32_t c = valid ? lead + big + extra_rep + tail : 0u;
				const uint32_t incl = wave_incl_scan(c);
				uint32_t o = ni + incl - c;
				if (valid) {
					if (lead)
						items[o++] = (uint16_t)v;
					for (uint32_t j = 0; j < big; j++)
						items[o++] = (uint16_t)(rep | ((full - base_len) << 8));
					if (extra_rep)
						items[o++] = (uint16_t)(extra_sym | ((rest - extra_base) << 8));
					for (uint32_t j = 0; j < tail; j++)
						items[o++] = (uint16_t)v;
					if (lead + tail)
						atomicAdd(&L.pfreq[v], lead + tail);
					if (big)
						atomicAdd(&L.pfreq[rep], big);
					if (extra_rep)
						atomicAdd(&L.pfreq[extra_sym], 1u);
				}
				ni += readlane(incl, 63);
			}
			if (lane == 0) {
				L.misc[0] = ni;
				L.misc[1] = hlit;
				L.misc[2] = hdist;
			}
			EMIT_T(1);
			build_code(L.pfreq, 19, HD_PRECODE_MAXBITS, L.pcode, Bd.hs, lane);
			uint32_t hclen = 19;
			while (hclen > 4 && (uniform(L.pcode[k_perm19[hclen - 1]]) >> 16) == 0)
				hclen--;
			// exact bit costs (extra bits are common to both codes)
			uint32_t dyn = 0, sta = 0, extra = 0;
			for (uint32_t base = 0; base < ni; base += 64) {
				const uint32_t k = base + lane;
				if (k < ni) {
					const uint32_t sym = Bd.items()[k] & 31;
					dyn += (L.pcode[sym] >> 16) + (sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u);
				}
			}
			for (uint32_t base = 0; base < 286; base += 64) {
				const uint32_t s = base + lane;
				if (s < 286) {
					const uint32_t f = L.lf[s];
					dyn += f * (Bd.lcode[s] >> 16);
					sta += f * (s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u);
					if (s >= 265 && s < 285)
						extra += f * ((s - 261) >> 2);
				}
			}
			if (lane < 30) {
				const uint32_t f = L.df[lane];
				dyn += f * (Bd.dcode[lane] >> 16);
				sta += f * 5u;
				extra += f * (lane < 4 ? 0u : (lane >> 1) - 1);
			}
			dyn = readlane(wave_incl_scan(dyn), 63) + 3 + 5 + 5 + 4 + 3 * hclen;
			sta = readlane(wave_incl_scan(sta), 63) + 3;
			extra = readlane(wave_incl_scan(extra), 63);
			const bool use_dynamic = dyn < sta;
			const uint64_t blockbits = (uint64_t)(use_dynamic ? dyn : sta) + extra;
			if ((uint64_t)(bitpos - paybase) + blockbits > limit_bits)
				return false;
			if (use_dynamic) {
				// BFINAL, BTYPE=10, HLIT, HDIST, HCLEN in lanes 0..4, then the precode lengths
				uint32_t c0 = 0, n0 = 0;
				if (lane == 0) { c0 = (final && !flush) ? 1u : 0u; n0 = 1; }
				else if (lane == 1) { c0 = 2; n0 = 2; }
				else if (lane == 2) { c0 = hlit - 257; n0 = 5; }
				else if (lane == 3) { c0 = hdist - 1; n0 = 5; }
				else if (lane == 4) { c0 = hclen - 4; n0 = 4; }
				else if (lane < 5 + hclen) { c0 = L.pcode[k_perm19[lane - 5]] >> 16; n0 = 3; }
				emit1(c0, n0);
				for (uint32_t base = 0; base < ni; base += 64) {
					const uint32_t k = base + lane;
					uint32_t cc = 0, nn = 0;
					if (k < ni) {
						const uint32_t it = Bd.items()[k], sym = it & 31;
						const uint32_t pc = L.pcode[sym];
						cc = (pc & 0xffff) | ((it >> 8) << (pc >> 16));
						nn = (pc >> 16) + (sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u);
					}
					emit1(cc, nn);
				}
			} else {
				emit1(lane == 0 ? ((final && !flush) ? 1u : 0u) : 1u, lane == 0 ? 1u : lane == 1 ? 2u : 0u);
				// static code tables in the same format
				for (uint32_t s = lane; s < 288; s += 64) {
					const uint32_t len = s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u;
					const uint32_t cw = s < 144 ? 0x30 + s : s < 256 ? 0x190 + (s - 144) : s < 280 ? s - 256 : 0xC0 + (s - 280);
					Bd.lcode[s] = (len << 16) | (__brev(cw) >> (32 - len));
				}
				if (lane < 32)
					Bd.dcode[lane] = (5u << 16) | (__brev(lane) >> 27);
			}
			// ---- pass 2: the tokens -------------------------------------------
			// straight-line: both forms are computed and one is selected; a token's two fields
			// (litlen code + extra bits, offset code + extra bits: <= 20 + 28 bits) go out as
			// one 64-bit OR over up to three dwords
			// The emit-only kernels code the litlen half of a token with ONE lookup: after the header is out (the code
			// construction's scratch is dead from there on) the block's code goes into a 512-entry table -- a literal's
			// codeword, or a length's with its extra bits behind it, and the bit count in the top byte -- as the level-1
			// kernel has it for the static code (CrcTables::SL).  Slot arithmetic + two selects per token before: ~17 of
			// the pass's ~75 vector instructions.
			uint32_t *const litlen_lut = Bd.hs.freq;             // 512 dwords: freq[288] and the first 224 of nf
			static_assert(offsetof(HuffScratch, nf) == sizeof(uint32_t) * 288, "the table runs from freq into nf");
			if (EMIT) {
#pragma unroll
				for (uint32_t q = 0; q < 4; q++) {
					const uint32_t i = 64 * q + lane;                // literal i, length 3 + i
					const uint32_t lc = Bd.lcode[i];
					litlen_lut[i] = (lc & 0xffff) | (lc >> 16 << 24);
					uint32_t ls, leb, lev;
					len_slot(i + 3, ls, leb, lev);
					const uint32_t mc = Bd.lcode[257 + ls];
					litlen_lut[256 + i] = (mc & 0xffff) | (lev << (mc >> 16)) | (((mc >> 16) + leb) << 24);
				}
			}
			auto put_tokens_n = [&](uint32_t tk, uint32_t nvalid) {
				const bool valid = lane < nvalid;
				const bool is_match = (tk & HD_TOKEN_MATCH) != 0;
				uint32_t ca, na;
				if (EMIT) {
					// (the workgroup parse's tokens carry the table index in bits 16..24; the others a literal in the low byte)
					const uint32_t idx = (!PARTS && a.wg) ? (tk >> 16) & 0x1ffu : is_match ? 256u + ((tk >> 16) & 0xffu) : tk & 0xffu;
					const uint32_t le = litlen_lut[idx];
					ca = le & 0xffffffu;
					na = le >> 24;
				} else {
					uint32_t ls, leb, lev;
					len_slot(((tk >> 16) & 0xff) + 3, ls, leb, lev);
					const uint32_t lc = Bd.lcode[is_match ? 257 + ls : (tk & 0xff)];
					ca = (lc & 0xffff) | (is_match ? lev << (lc >> 16) : 0u);
					na = (lc >> 16) + (is_match ? leb : 0u);
				}
				uint32_t ds, deb, dev;
				off_slot((tk & 0xffff) + 1, ds, deb, dev);
				const uint32_t dc = Bd.dcode[ds];
				const uint32_t cb = (dc & 0xffff) | (dev << (dc >> 16));
				const uint32_t nb = is_match ? (dc >> 16) + deb : 0u;
				const uint32_t nbits = valid ? na + nb : 0u;
				const uint64_t code = valid ? ((uint64_t)ca | ((uint64_t)(is_match ? cb : 0u) << na)) : 0ull;
				const uint32_t incl = wave_incl_scan(nbits);
				const uint32_t at = bitpos + incl - nbits;
				const uint32_t sh = at & 31, i = (at >> 5) & (STG - 1);
				const uint64_t lo = code << sh;                      // bits [0, 64) of the shifted field
				const uint32_t hi = sh ? (uint32_t)(code >> (64 - sh)) : 0u;   // and what a 48-bit field spills beyond
				atomicOr(&stage[i], (uint32_t)lo);
				atomicOr(&stage[(i + 1) & (STG - 1)], (uint32_t)(lo >> 32));
				atomicOr(&stage[(i + 2) & (STG - 1)], hi);
				bitpos += readlane(incl, 63);
				flush_ready();
			};
			auto put_tokens = [&](uint32_t tk, uint32_t base) { put_tokens_n(tk, ntok_slab - base); };
			EMIT_T(2);
			if (EMIT && !PARTS && a.wg) {
				wg_for_tokens(put_tokens_n);
			} else if (EMIT) {
				// The emit-only kernel reads its tokens from HBM: one dependent load per 64 tokens left the
				// wave waiting for memory most of the time (~0.5 us per iteration).  Tokens come in groups of
				// eight loads issued back to back, and the next group is requested before the current one is
				// coded.  (Reloading a register right after its use does not work: with the staging ring's
				// stores in the loop the compiler waits for vmcnt(0) before every use, i.e. for the load just
				// issued as well.)
				constexpr uint32_t PF = 8;
				uint32_t cur[PF], nxt[PF];
#pragma unroll
				for (uint32_t j = 0; j < PF; j++)
					nxt[j] = 64 * j + lane < ntok_slab ? tok[tix(64 * j + lane)] : 0u;
				for (uint32_t base = 0; base < ntok_slab; base += 64 * PF) {
#pragma unroll
					for (uint32_t j = 0; j < PF; j++) {
						cur[j] = nxt[j];
						const uint32_t kn = base + 64 * (PF + j) + lane;
						nxt[j] = kn < ntok_slab ? tok[tix(kn)] : 0u;
					}
#pragma unroll
					for (uint32_t j = 0; j < PF; j++)
						if (base + 64 * j < ntok_slab)
							put_tokens(cur[j], base + 64 * j);
				}
			} else {
				for (uint32_t base = 0; base < ntok_slab; base += 64)
					put_tokens(base + lane < ntok_slab ? tok[base + lane] : 0u, base);
			}
			EMIT_T(3);
			{
				const uint32_t eob = Bd.lcode[256];
				emit1(lane == 0 ? (eob & 0xffff) : 0u, lane == 0 ? (eob >> 16) : 0u);
			}
			for (uint32_t i = lane; i < 288; i += 64)
				L.lf[i] = 0;
			if (lane < 32)
				L.df[lane] = 0;
			ntok = 0;
			ntok_slab = 0;
			return true;
		};

		// queued tokens -> slab (one coalesced 4 B/lane store) + symbol histograms (ds_add_u32)
		auto drain_tokens = [&](uint32_t count) {
			const uint32_t t = tokq[(qhead + lane) & (TOKQ - 1)];
			qhead += count;
			if (lane < count) {
				tok[ntok_slab + lane] = t;
				if (t & HD_TOKEN_MATCH) {
					uint32_t ls, leb, lev, ds, deb, dev;
					len_slot(((t >> 16) & 0xff) + 3, ls, leb, lev);
					off_slot((t & 0xffff) + 1, ds, deb, dev);
					atomicAdd(&L.lf[257 + ls], 1u);
					atomicAdd(&L.df[ds], 1u);
				} else {
					atomicAdd(&L.lf[t & 0xff], 1u);
				}
			}
			ntok_slab += count;
		};

		uint32_t crcv;
		if constexpr (!EMIT) {
		// ---- pass 1: the parse ------------------------------------------------
		Fetched f0 = { 0, 0, 0, 0 }, f1 = { 0, 0, 0, 0 };
		Probed q0 = { 0, 0, 0 };
		if (alive && n) {
			fill_piece();
			f0 = fetch(0);
			q0 = probe(f0.c);
			f1 = fetch(64);
		}
		uint32_t carry = 0;
		for (uint32_t S = 0; S < n && alive; S += 64) {
			if (filled < n && filled < S + HD_LOOKAHEAD)
				fill_piece();
			const uint32_t lo = filled > W ? filled - W : 0;
			const uint32_t lanes = n - S < 64 ? n - S : 64;
			const Fetched fc = f0;
			const Probed qc = q0;
			f0 = f1;
			q0 = probe(f1.c);
			f1 = fetch(S + 128);

			const uint32_t p = S + lane;
			const bool can = p + (DEEP ? HD_LAZY_KEY_BYTES : HD_MIN_MATCH) <= n;
			const uint32_t cv0 = fc.v, cvh0 = fc.vh;
			const uint32_t room = n - p;
			uint32_t cp = fc.c - 1;
			bool had = can && fc.c != 0 && cp >= lo;
			uint32_t cv = __builtin_amdgcn_alignbyte(qc.c1, qc.c0, cp & 3);
			uint32_t cvh = __builtin_amdgcn_alignbyte(qc.c2, qc.c1, cp & 3);
			if (DEEP) {
				// both positions of the bucket, verified over 16 bytes (read here, byte by byte: this kernel only takes
				// the blocks the split path leaves -- its speed does not matter, its bytes must be the twin's);
				// the older one is taken only when it is strictly longer
				auto prefix16 = [&](uint32_t cpos) -> uint32_t {
					const uint32_t lim = room < 16 ? room : 16;
					uint32_t k = 0;
					while (k < lim && ring8[(p + k) & (W - 1)] == ring8[(cpos + k) & (W - 1)])
						k++;
					return k;
				};
				const uint32_t cpB = fc.c2 - 1;
				const bool hadB = can && fc.c2 != 0 && cpB >= lo;
				const uint32_t LA = had ? prefix16(cp) : 0u, LB = hadB ? prefix16(cpB) : 0u;
				const uint32_t LAv = LA >= 4 ? LA : 0u, LBv = LB >= 4 ? LB : 0u;
				if (LBv > LAv) {
					cp = cpB;
					had = hadB;
					const uint32_t *wb = &ring32[(cpB >> 2) & W4M];
					cv = __builtin_amdgcn_alignbyte(wb[1], wb[0], cpB & 3);
					cvh = __builtin_amdgcn_alignbyte(wb[2], wb[1], cpB & 3);
				}
			}
			const uint32_t x = cvh ^ cvh0;
			const uint32_t eqb = x ? (uint32_t)(__ffs((int)x) - 1) >> 3 : 4u;
			uint32_t mylen = 4 + eqb < room ? 4 + eqb : room;
			bool ok = had && cv == cv0 && mylen >= (uint32_t)MINLEN;
			if (LAZY) {
				// a candidate steps aside when its right neighbour's 8-byte length is longer
				const uint32_t l8 = ok ? (mylen < 8 ? mylen : 8u) : 0u;
				const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)l8, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
				const bool defer = ok && lane + 1 < lanes && nx > l8;
				ok = ok && !defer;
			}
			const uint32_t dist = ok ? p - cp : 1u;

			if (carry >= lanes) {
				carry -= lanes;
			} else {
				const bool capped = ok && eqb == 4 && room > 8;
				const uint32_t jump8 = ok ? (mylen < 8 ? mylen : 8u) : 1u;
				uint64_t starts;
				{
					const Fn8 w = fn8_scan(fn8_make(lane >= carry, jump8 - 1));
					const uint32_t sin = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(w.lo & 0xff), 0x138, 0xf, 0xf, false);
					starts = __ballot(sin == 0 && lane >= carry);
				}
				const uint64_t capmask = __ballot(capped);
				uint64_t cm = starts & capmask;
				while (cm) {
					const uint32_t m = (uint32_t)__ffsll((unsigned long long)cm) - 1;
					const uint32_t dm = readlane(dist, m);
					const uint32_t pm = S + m;
					const uint32_t maxlen = n - pm < HD_MAX_MATCH ? n - pm : HD_MAX_MATCH;
					uint32_t len = 8;
					for (;;) {
						const uint32_t idx = len + lane;
						bool diff = true;
						if (idx < maxlen)
							diff = ring8[(pm + idx) & (W - 1)] != ring8[(pm + idx - dm) & (W - 1)];
						const uint64_t nq = __ballot(diff);
						const uint32_t k = nq ? (uint32_t)__ffsll((unsigned long long)nq) - 1 : 64;
						len += k;
						if (k < 64)
							break;
					}
					const uint64_t upto_m = (2ull << m) - 1;
					if (len > 8) {
						if (lane == m)
							mylen = len;
						const uint32_t q = m + len;
						uint64_t fresh = 0;
						uint32_t xx = q;
						while (xx < 64 && !((starts >> xx) & 1)) {
							fresh |= 1ull << xx;
							xx += readlane(jump8, xx);
						}
						const uint64_t below_x = xx >= 64 ? ~0ull : ((1ull << xx) - 1);
						starts = (starts & (upto_m | ~below_x)) | fresh;
					}
					cm = starts & capmask & ~upto_m;
				}
				const uint32_t last = 63 - (uint32_t)__clzll((long long)starts);
				const uint32_t E = last + readlane(ok ? mylen : 1u, last);
				const bool is_start = (starts >> lane) & 1;
				const bool is_match = is_start && ok;
				const bool is_lit = is_start && !ok && lane < lanes;
				carry = (lanes == 64 && E > 64) ? E - 64 : 0;

				// tokens -> LDS queue (compacted in lane order); the slab store and the
				// histogram updates cost the same for 1 or 64 tokens, so they wait for 64
				const bool is_tok = is_match || is_lit;
				const uint64_t tm = __ballot(is_tok);
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32),
										 __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0));
				if (is_tok)
					tokq[(qtail + rank) & (TOKQ - 1)] =
						is_match ? (HD_TOKEN_MATCH | ((mylen - 3) << 16) | (dist - 1)) : (cv0 & 0xff);
				qtail += (uint32_t)__popcll(tm);
				if (qtail - qhead >= 64)
					drain_tokens(64);
			}
			ntok = ntok_slab + (qtail - qhead);          // tokens of the open block, queued ones included
			if (ntok >= HD_DYN_BLOCK_TOKENS && S + 64 < n) {
				if (qtail != qhead)
					drain_tokens(qtail - qhead);
				// our own token stores must be visible to our own loads
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				alive = flush_block(false);
				// the code construction used the ring's LDS: bring the window back
				if (alive) {
					const uint32_t first = filled > W ? (filled - W) / HD_PIECE : 0;
					for (uint32_t piece = first; piece * HD_PIECE < filled; piece++) {
						const uint4 v = load_slot(src, n, piece, lane, aligned);
						const uint32_t ro = (piece * HD_PIECE) & (W - 1);
						((uint4 *)ring32)[ro / 16 + lane] = v;
						if (ro == 0 && lane == 0)
							((uint4 *)ring32)[W / 16] = v;
					}
				}
			}
		}
		if (alive) {
			if (qtail != qhead)
				drain_tokens(qtail - qhead);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			alive = flush_block(true);
		}

		while (filled < n) {
			const uint32_t piece = filled / HD_PIECE;
			const uint4 pv = pre;
			filled += HD_PIECE;
			if (filled < n)
				pre = load_slot(src, n, piece + 1, lane, aligned);
			crc.template fold<true>(ct, piece, piece * HD_PIECE + lane * 16 + 16 <= n, pv);
		}
		crcv = crc.finish(ct, lane, n, src + (n & ~15u));
		} else {
			const uint32_t bi = (BESIDE && a.span_sub) ? b % a.span_sub : b - a.first;
			if (PARTS) {
				// the parse has been done in parts, one wavefront each (k_deflate_static, `parted`): ONE DEFLATE block
				// of all their tokens, its histograms the sums, its CRC from crc(A || B) = crc(A) x^(8 |B|) ^ crc(B)
				const SplitLayout lay = part_layout();
				const uint32_t rec_dw = (uint32_t)(lay.bytes / 4);
				const uint8_t *rec0 = part_block(a.scratch, bi * PARTS);
				const uint32_t np = (n + HD_LAT_PART_BYTES - 1) / HD_LAT_PART_BYTES;      // <= PARTS (the launch's promise)
				uint32_t lf[5] = { 0, 0, 0, 0, 0 }, dfv = 0, total = 0, prev = 0;
#pragma unroll
				for (int q = 0; q < PARTS; q++) {
					const uint8_t *rec = rec0 + (uint64_t)q * lay.bytes;
					const bool there = (uint32_t)q < np;
					const uint32_t *h = (const uint32_t *)(rec + lay.off_hist);
#pragma unroll
					for (int i = 0; i < 5; i++)
						lf[i] += (there && 64 * i + lane < 288) ? h[64 * i + lane] : 0u;
					dfv += (there && lane < 32) ? h[288 + lane] : 0u;
					const uint32_t ntq = there ? uniform(*(const uint32_t *)(rec + lay.off_ntok)) : 0u;
					pcum[q] = total;
					pgap[q] = q ? rec_dw - prev : 0u;
					total += ntq;
					prev = ntq;
				}
#pragma unroll
				for (int i = 0; i < 5; i++)
					if (64 * i + lane < 288)
						L.lf[64 * i + lane] = lf[i];
				if (lane < 32)
					L.df[lane] = dfv;
				tok = (uint32_t *)rec0;
				ntok_slab = total;
				alive = flush_block(true);
				crcv = uniform(L.misc[4]);              // (the second wavefront's, behind the barriers of flush_block)
			} else if (a.wg) {
				// the workgroup parse has left the pieces' tokens and counts: the DEFLATE blocks are cut here -- at a piece
				// boundary once a block holds HD_DYN_BLOCK_TOKENS tokens or when the token mix has shifted (the twin's
				// wg_split_check; deflate_compress.c:2141-2218) --, their symbols counted, and each is closed as ever
				const SplitLayout lay = wg_layout(a.split_max);
				const uint8_t *rec = ((BESIDE && a.span_sub && ((b / a.span_sub) & 1)) ? a.scratch_b : a.scratch) + (uint64_t)bi * lay.bytes;
				const uint32_t *m = (const uint32_t *)(rec + lay.off_rec);
				bool waited_out = false;
				if constexpr (BESIDE != 0) {
					// beside the parse (hd_deflate_wg.hpp launch_wg): the block's workgroup says when its records are complete.  The
					// polls are RMWs (they execute at the device's coherence point: no XCD's L2 can answer them with an old copy), a
					// flag per 128-byte line, ~60 us apart; behind the wait an acquire at agent scope makes this XCD's L2 forget its
					// clean lines, and the records -- written through their XCD's L2 by the parse (sc1) -- are read from memory.
					// Bounded (~2 s where a whole sub-batch's parse is tens of milliseconds): stored and counted, never a hang
					uint32_t spins = 0;
					while (__hip_atomic_fetch_add(&a.ready[32 * (size_t)(a.span_sub ? b : bi)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
						if (++spins > (1u << 15)) {
							waited_out = true;
							break;
						}
#pragma unroll
						for (int z = 0; z < 16; z++)
							__builtin_amdgcn_s_sleep(127);
					}
#if HD_BESIDE_SC1_LOADS
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // (order only: the records are read by sc1 loads)
#else
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
					if (waited_out && a.stalls && lane == 0)
						atomicAdd(a.stalls, 1u);
				}
				wg_pieces = (const uint4 *)(rec + lay.off_ntok);
				wg_base = 0xffffff00u;
				tok = (uint32_t *)rec;
				uint32_t m0;
				if (BESIDE && HD_BESIDE_SC1_LOADS) {
					m0 = __hip_atomic_load((hd_global_u32p)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					crcv = __hip_atomic_load((hd_global_u32p)m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				} else {
					m0 = m[0];
					crcv = m[1];
				}
				if (m0 != 0 || waited_out) {
					alive = false;
					if (n <= a.split_max)                        // (given up, not refused: the member is written stored, with the CRC of all of it)
						crcv = crc_of_block(ct, src, n, lane);
				}
				// (a refused block is longer than the slot its record was sized by: its pieces' counts are not there to be
				// walked -- ADVICE r4: the walk below ran over n / 1024 records before it looked at `alive`)
				wg_np = alive ? (n + HD_WG_CUT - 1) / HD_WG_CUT : 0u;
				for (uint32_t i = lane; i < 288; i += 64)
					L.lf[i] = 0;
				if (lane < 32)
					L.df[lane] = 0;
				uint32_t k = 0, block_begin = 0;
				do {
					uint32_t obs0 = 0, obs1 = 0, obs2 = 0, sn = 0, no0 = 0, no1 = 0, no2 = 0, snn = 0, blk_tok = 0;
					bool end = false;
					wg_k0 = k;
					EMIT_T0();
					while (k < wg_np && !end) {
						wg_piece_load(k);
						const uint32_t li = k - wg_base;
						blk_tok += readlane(wg_pv.x, li);
						snn += readlane(wg_pv.x, li);
						no0 += readlane(wg_pv.y, li);
						no1 += readlane(wg_pv.z, li);
						no2 += readlane(wg_pv.w, li);
						k++;
						const uint32_t here = k * HD_WG_CUT;             // (< n unless this was the last piece)
						if (k < wg_np) {
							end = blk_tok >= HD_DYN_BLOCK_TOKENS;
							if (!end && snn >= HD_WG_SPLIT_OBS && here - block_begin >= HD_WG_SPLIT_MIN && n - here >= HD_WG_SPLIT_MIN) {
								if (sn > 0) {
									const uint32_t e0 = obs0 * snn, a0 = no0 * sn, e1 = obs1 * snn, a1 = no1 * sn, e2 = obs2 * snn, a2 = no2 * sn;
									const uint32_t total = (a0 > e0 ? a0 - e0 : e0 - a0) + (a1 > e1 ? a1 - e1 : e1 - a1) + (a2 > e2 ? a2 - e2 : e2 - a2);
									const uint32_t items = sn + snn, blen = here - block_begin;
									uint32_t cutoff = snn * 200u / 512u * sn;
									if (blen < 10000 && items < 8192)
										cutoff += (cutoff >> 13) * (8192u - items);
									end = total + (blen / 4096u) * sn >= cutoff;
								}
								if (!end) {
									obs0 += no0;
									obs1 += no1;
									obs2 += no2;
									sn += snn;
									no0 = no1 = no2 = snn = 0;
								}
							}
						}
					}
					wg_k1 = k;
					block_begin = k * HD_WG_CUT;
					EMIT_T(4);
					if (!alive)
						break;
					// the block's symbols: the litlen symbol of a token is one lookup (length symbols from a 256-byte table that
				// is set up here, in scratch the code construction overwrites), the offset symbol slot arithmetic
				{
					uint8_t *const lsym = (uint8_t *)&Bd.hs.nf[256];
					uint32_t w = 0;
#pragma unroll
					for (uint32_t q = 0; q < 4; q++) {
						uint32_t ls, eb, ev;
						len_slot(4 * lane + q + 3, ls, eb, ev);
						w |= ls << (8 * q);
					}
					((uint32_t *)lsym)[lane] = w;
#if HD_EXP_NO_COUNT != 2                      /* experiment (tools/r05_nocount.sh, timing only: with no counts every block gets the static code -- valid, larger) */
					wg_for_tokens([&](uint32_t tk, uint32_t nv) {
						const bool is_match = (tk & HD_TOKEN_MATCH) != 0;
						const uint32_t idx = (tk >> 16) & 0x1ffu;                // literal, or 256 + (length - 3)
						const uint32_t sym = is_match ? 257u + lsym[idx & 0xffu] : idx;
						uint32_t ds, eb, ev;
						off_slot((tk & 0xffff) + 1, ds, eb, ev);
#if HD_EXP_NO_COUNT == 1
						if (lane < nv && sym + ds == 0xffffffffu)     // (never: the arithmetic stays, the atomics go)
							atomicAdd(&L.lf[0], 1u);
#else
						if (lane < nv) {
							atomicAdd(&L.lf[sym], 1u);
							if (is_match)
								atomicAdd(&L.df[ds], 1u);
						}
#endif
					});
#endif
				}
					EMIT_T(5);
					ntok_slab = blk_tok;
					alive = flush_block(k == wg_np);
					EMIT_T(6);
				} while (alive && k < wg_np);
			} else {
			// the parse has been done: one flush per recorded DEFLATE block
			const SplitLayout lay = split_layout(a.split_max);
			const uint8_t *rec = a.scratch + (uint64_t)bi * lay.bytes;
			const uint32_t *m = (const uint32_t *)(rec + lay.off_rec);
			const uint32_t *nt = (const uint32_t *)(rec + lay.off_ntok);
			uint32_t ndb = m[0];
			crcv = m[1];
			uint32_t t0 = 0;
			for (uint32_t k = 0; k < ndb && alive; k++) {
				const uint32_t *h = (const uint32_t *)(rec + lay.off_hist) + k * 320;
				for (uint32_t i = lane; i < 288; i += 64)
					L.lf[i] = h[i];
				if (lane < 32)
					L.df[lane] = h[288 + lane];
				tok = (uint32_t *)rec + t0;
				ntok_slab = nt[k];
				alive = flush_block(k + 1 == ndb);
				t0 += nt[k];
			}
			}
		}

		if (!alive) {
			write_stored_member(a, b, src, n, crcv, lane);
			continue;
		}
		if (flush) {
			// empty stored block header (000), alignment, LEN = 0, NLEN = ffff
			bitpos = (bitpos + 3 + 7) & ~7u;
			emit1(lane == 1 ? 0xffffu : 0u, lane < 2 ? 16u : 0u);
		}
		bitpos = (bitpos + 7) & ~7u;
		const uint32_t paylen = (bitpos - paybase) >> 3;
		if (trl) {
			uint32_t tcode = 0, nb = 0;
			if (lane < trl / 2) {
				tcode = frame_trl_field(a.frame, lane, crcv, n);
				nb = 16;
			}
			emit1(tcode, nb);
		}
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
}

// Blocks up to a.split_max: parse kernel (the level-1 kernel with this level's parse parameters, at the
// occupancy its ring + table allow and without a persistent loop), tokens + histograms through HBM,
// then the one emit-only kernel (16 waves per CU).  Larger blocks (none, unless a block is larger
// than its slot and will fail anyway, or the scratch budget cannot hold even one): the fused kernel.
template <int W, int H, int MINLEN, int LAZY, int INTRA, int DEEP = 0>
inline void launch_level(const DeflateArgs &a, int level, hipStream_t st)
{
	const uint32_t sub = a.parts ? part_sub_batch(a.nblocks, a.parts) : split_sub_batch(a.nblocks, a.split_max, level);
	DeflateArgs s = a;
	// scratch: [ fused slabs | overflow flags, one u32 per block | split records of one sub-batch ]
	s.split_ovf = (uint32_t *)(a.scratch + fused_scratch_bytes(a.nblocks, level));
	s.scratch = (uint8_t *)s.split_ovf + (((uint64_t)a.nblocks * 4 + 15) & ~(uint64_t)15);
	if (a.parts) {
		// latency segments (hd_segment.hpp sets a.parts = HD_LAT_PARTS): a parse wavefront per part, an emit wavefront per
		// segment, and nothing for a fused kernel to take over -- a part's slab holds a token per byte
		for (uint32_t first = 0; first < a.nblocks; first += sub) {
			s.first = first;
			s.count = a.nblocks - first < sub ? a.nblocks - first : sub;
			hipLaunchKernelGGL((k_deflate_static<W, H, true, MINLEN, LAZY, INTRA, DEEP>), dim3(s.count * HD_LAT_PARTS_MAX), dim3(64), 0, st, s);
			const uint32_t eg = s.count < 256u * 16u ? s.count : 256u * 16u;
			hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1, 0, 0, HD_LAT_PARTS_MAX>), dim3(eg),
					   dim3(128), 0, st, s);                 // (two wavefronts per workgroup: see the kernel)
		}
		return;
	}
	for (uint32_t first = 0; sub && first < a.nblocks; first += sub) {
		s.first = first;
		s.count = a.nblocks - first < sub ? a.nblocks - first : sub;
		hipLaunchKernelGGL((k_deflate_static<W, H, true, MINLEN, LAZY, INTRA, DEEP>), dim3(s.count), dim3(64), 0, st, s);
		const uint32_t eg = s.count < 256u * 16u ? s.count : 256u * 16u;
		hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1>), dim3(eg), dim3(64), 0,
				   st, s);
	}
	DeflateArgs f = a;
	f.split_ovf = s.split_ovf;
	f.skip_small = sub ? 1 : 0;                          // nothing went the split way: the fused kernel takes all
	hipLaunchKernelGGL((k_deflate_dynamic<W, H, MINLEN, LAZY, 0, INTRA, DEEP>), dim3(dynamic_grid(a.nblocks, level)), dim3(64), 0, st, f);
}

void launch_wg(const DeflateArgs &a, int level, hipStream_t st);       // hd_deflate_wg.hpp

inline int launch_deflate_dynamic(const DeflateArgs &a, int level, hipStream_t st)
{
	// the workgroup levels: one workgroup per block, then the emit-only kernel (the one-wavefront one, or -- a.lat, the
	// per-block boundary -- a workgroup per member: hd_emit_wg.hpp).  Since round 5 these levels have no other form: the
	// latency segments parsed in parts (a.parts) are level 2's alone
	if (level >= HD_WG_LEVEL) {
		launch_wg(a, level, st);
		return 0;
	}
	launch_level<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, HD_INTRA_DIST>(a, level, st);
	return 0;
}

} // namespace hd
